#!/bin/bash
# GPU timeline of the bench loop: average kernel durations and the idle gap in front of each kernel (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/gaps
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/err
python - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f))))
# keep the steady-state part: last 60%
rows = rows[int(len(rows) * 0.4):]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    dur[n1].append(e1 - s1); gap[n1].append(s1 - e0)
tot = 0
for n in dur:
    if len(dur[n]) < 10: continue
    d = sum(dur[n]) / len(dur[n]) / 1e3; g = sum(gap[n]) / len(gap[n]) / 1e3
    print('%-42s n=%4d dur %7.2f us  gap-before %7.2f us' % (n, len(dur[n]), d, g))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.json
