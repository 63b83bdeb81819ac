#!/bin/bash
# dev: per-kernel time of one kernel (name fragment $1) in the deformation bench, for each variant library under 4dgs-slam_amd/_variants
cd /tmp && export TMPDIR=/tmp
for lib in /root/repo/4dgs-slam_amd/_variants/*.so; do
  O=/root/repo/gpurun_out/var_$(basename $lib .so); rm -rf $O; mkdir -p $O
  GSR_GLUE=ctypes GSR_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python /root/repo/tools/bench_deformation.py --n ${2:-200000} --iters 10 --only-network > $O/out.json 2> $O/err
  echo "== $(basename $lib) $(tail -1 $O/out.json | grep -o 'fused_field": [0-9.]*')"
  python - $O "$1" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Name']:
        print('   %9.1f us  %s' % (float(r['AverageNs']) / 1e3, r['Name'][:70]))
PY
done
