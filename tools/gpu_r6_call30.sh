#!/bin/bash
# whole-run kernel census of the SLAM demo scenarios
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/democensus; rm -rf $O; mkdir -p $O
for sc in static_640x480_graph dynamic_640x480_graph; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$sc -o s -- python $R/tools/run_slam_demo.py --only $sc > $O/$sc.json 2> $O/$sc.err
python - $O/$sc <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(sys.argv[1].split('/')[-1], "device busy %.3f s, %d kernels launched" % (tot / 1e9, sum(int(r['Calls']) for r in rows)))
for r in rows[:22]:
    print('%8d calls %8.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:100]))
PY
rm -rf $O/$sc
done
