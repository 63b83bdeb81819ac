#!/bin/bash
# dev: the tracking iteration after a change -- its tests, tools/bench_tracking.py, and the per-kernel times of the 10 k-Gaussian map
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06; mkdir -p $O
( cd $R && timeout 900 python -m pytest tests/test_hip_slam.py -x -q -m gpu -k "track or latch or camera_step" -p no:cacheprovider 2>&1 | grep -v Warning | tail -15 )
python $R/tools/bench_tracking.py 2>/dev/null | tail -1 > $O/tracking_graph.json; cat $O/tracking_graph.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trk_stats -o t -- python $R/tools/bench_tracking.py 10000 > /dev/null 2> $O/trk.err
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trk_stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print('%8d calls %9.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:90]))
PY
rm -rf $O/trk_stats
