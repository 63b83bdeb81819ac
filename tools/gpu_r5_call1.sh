#!/bin/bash
# round 5, first GPU call: the batched-time deformation path (tests, config #3 timing, kernel breakdown)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c1; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_deformation.py -x -q -k "batched or render_views_dynamic or golden or render_dynamic" 2>&1 | tail -15
timeout 600 python tools/bench_config3.py --modes fused,batched --iters 5 > $O/config3.json 2> $O/config3.err; tail -3 $O/config3.err; cat $O/config3.json
R=/root/repo
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/tools/bench_config3.py --modes batched --iters 3 > $R/$O/prof_out.json 2> $R/$O/prof_err )
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5c1/prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:30]:
    print('%6.2f%% %6d calls %9.1f us avg  %s' % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
import shutil; shutil.copy(f, 'gpurun_out/r5c1/config3_kernel_stats.csv')
PY
rm -rf $O/prof
