cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c5; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/out.json 2> $O/err.txt
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/st/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("device busy %.3f s" % (tot / 1e9))
for r in rows[:40]:
    print('%8d calls %9.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:100]))
PY
rm -rf $O/st
