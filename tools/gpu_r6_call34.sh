cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pv; rm -rf $O; mkdir -p $O
for m in 0 1; do
GSR_HEX_ORDERED=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$m -o s -- python $R/tools/bench_config3.py --modes fused --iters 3 > $O/out$m.json 2> $O/err$m.txt
python - $O/st$m $m <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("ordered =", sys.argv[2])
for r in rows:
    if any(k in r['Name'] for k in ('hexsort', 'hexord', 'hexplane', 'fillBuffer', 'Memset')):
        print('%8d calls %9.2f us avg %6.2f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:90]))
PY
tail -c 200 $O/out$m.json; echo
rm -rf $O/st$m
done
