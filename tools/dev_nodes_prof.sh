#!/bin/bash
# dev: rocprof kernel stats of the control-node bench
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/nodeprof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python /root/repo/tools/bench_control_nodes.py --n ${1:-100000} --iters 10 > $O/out.json 2> $O/err
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/nodeprof/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'gsr::' in r['Name']]
for r in rows:
    print('%8d calls %9.1f us avg  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:100]))
PY
tail -1 $O/out.json
