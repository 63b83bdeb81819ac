#!/bin/bash
# dev: rocprof kernel stats of the deformation network fwd+bwd (where does the torch MLP spend its time?)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/mlpprof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python /root/repo/tools/bench_deformation.py --n 200000 --iters 10 --only-network > $O/out.json 2> $O/err
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/mlpprof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print('%6.2f%% %8d calls %9.1f us avg  %s' % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
cat $O/out.json | tail -1
