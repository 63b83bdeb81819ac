#!/bin/bash
# dev: per-kernel times of the tracking iteration at 10 k and 50 k Gaussians
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trk; rm -rf $O; mkdir -p $O
for n in 10000 50000; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$n -o t -- python $R/tools/bench_tracking.py $n > $O/out$n.txt 2> $O/err$n.txt
python - $O/s$n <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:18]:
    print('%8d calls %9.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:100]))
PY
tail -1 $O/out$n.txt
done
find $O -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' \) -delete
