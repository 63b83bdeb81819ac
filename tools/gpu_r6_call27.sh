#!/bin/bash
# where the device time of the whole reference-schedule stand-in run goes (rocprofv3 kernel stats over the run)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/standin; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python $R/tools/run_config4_stand_in.py > $O/out.json 2> $O/err.txt
python - $O <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + '/st/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("device busy %.2f s, %d kernels launched" % (tot / 1e9, sum(int(r['Calls']) for r in rows)))
for r in rows[:45]:
    print('%8d calls %8.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:100]))
d = json.load(open(sys.argv[1] + '/out.json')); print(d['seconds'], d['fps'])
PY
cp $(find $O/st -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/st
