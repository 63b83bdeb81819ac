#!/bin/bash
# dev: fabric traffic per launch of the step's kernels (FETCH_SIZE / WRITE_SIZE in separate PMC passes, gfx950 half-count correction) for the
# library in GSR_LIB (default: the product library), one line per kernel
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/traffic_$$; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > /dev/null 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > /dev/null 2> $O/w.err
python - $O <<'PY'
import csv, glob, sys, collections
def agg(pat):
    d = collections.defaultdict(list)
    for p in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(p)):
            if "gsr::" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("gsr::")[1].split("(")[0].split("<")[0].replace("_kernel", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}
f, w = agg(sys.argv[1] + "/f/**/*counter_collection.csv"), agg(sys.argv[1] + "/w/**/*counter_collection.csv")
tot = 0
for k in sorted(f):
    b = (2 * f[k] + w.get(k, 0)) * 1024
    tot += b
    print("%-22s fetch %7.1f MB  write %7.1f MB  total %7.1f MB" % (k, 2 * f[k] / 1024, w.get(k, 0) / 1024, b / 1e6 * 1.048576 / 1.048576))
print("whole step %.1f MB" % (tot / 1e6))
PY
rm -rf $O
