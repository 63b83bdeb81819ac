#!/bin/bash
# round 5: config #3 tests at full size, its profile artefacts, the bench line with the config3 block
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_configs.py -x -q -s -k "config3" 2>&1 | grep -v "^$" | tail -6
timeout 900 python tools/bench_config3.py --iters 5 > $O/r05_config3.json 2> $O/config3.err; tail -2 $O/config3.err; cat $O/r05_config3.json
R=/root/repo
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/tools/bench_config3.py --modes batched --iters 3 > $R/$O/prof_out.json 2> $R/$O/prof_err )
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5c3/prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6, '(5 iterations: 2 warm-up + 3 timed)')
for r in rows[:24]:
    print('%6.2f%% %6d calls %9.1f us avg  %s' % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
import shutil; shutil.copy(f, 'gpurun_out/r5c3/r05_config3_kernel_stats.csv')
PY
rm -rf $O/prof
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c3/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('config3'), d.get('config5', {}).get('ms_per_step'))
PY
