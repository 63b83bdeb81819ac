#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/cfg3prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python /root/repo/tools/bench_config3.py --views 2 --iters 3 --fused-only > $O/out.json 2> $O/err
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/cfg3prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:32]:
    print('%6.2f%% %6d calls %9.1f us avg  %s' % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
tail -1 $O/out.json
