#!/bin/bash
# phase1_views load batching + views_reduce + tau_sum: tests and timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/p1; mkdir -p $O
timeout 300 python tools/dev_hexviews.py 2>/dev/null | tail -1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st -o c -- python $GRAFT_REPO_ROOT/tools/dev_hexviews.py > /dev/null 2> $GRAFT_REPO_ROOT/$O/err.txt )
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/p1/st/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print('%6d %9.1f us  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
rm -rf $O/st
if [ "$1" != "quick" ]; then
timeout 900 python -m pytest tests/test_hip_deformation.py tests/test_hip_views.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -4
timeout 900 python -m pytest tests/test_hip_configs.py -x -q -m gpu -p no:cacheprovider -k "config3" 2>&1 | grep -v Warning | tail -4
timeout 600 python tools/bench_config3.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if 'ms_per_iteration' in k})"
fi
