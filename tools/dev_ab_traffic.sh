#!/bin/bash
# dev: per-kernel time AND fabric traffic (FETCH_SIZE / WRITE_SIZE passes) of the default bench for each variant library under
# 4dgs-slam_amd/_variants/*.so, on ONE box. Prints MB per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, see collect_round2.py).
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for lib in $R/4dgs-slam_amd/_variants/*.so; do
  n=$(basename $lib .so); O=$R/gpurun_out/abt_$n; rm -rf $O; mkdir -p $O
  CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
  GSR_GLUE=ctypes GSR_LIB=$lib rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > /dev/null 2> $O/f.err
  GSR_GLUE=ctypes GSR_LIB=$lib rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > /dev/null 2> $O/w.err
  echo "== $n"
  python - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("f", "w"):
    for f in glob.glob(sys.argv[1] + f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].split("::")[-1][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for k, c in sorted(acc.items()):
    if not any(s in k for s in ("preprocess", "scatter", "render", "geometry", "sort", "tile_offsets", "scan")): continue
    fe = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"])); wr = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
    mb = (2 * fe + wr) * 1024 / 1e6; tot += mb
    print("   %-28s fetch %7.1f MB  write %6.1f MB  total %7.1f MB" % (k, 2 * fe * 1024 / 1e6, wr * 1024 / 1e6, mb))
print("   sum %.1f MB" % tot)
PY
  find $O -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' \) -delete
done
