#!/bin/bash
mkdir -p gpurun_out
{
for rep in 1 2; do
for cfg in "0 1" "1 0" "1 1"; do
  set -- $cfg
  echo "GSR_DENSE_TRUNK=$1 GSR_DENSE_WIDE=$2"
  GSR_DENSE_TRUNK=$1 GSR_DENSE_WIDE=$2 python tools/mapping_iteration_launches.py --dynamic 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
print({k:d[k] for k in ('device_us_per_iteration',)}, d['graph']['ms_per_iteration_without_capture'], d['graph']['second_call'])
"
done
done
} > gpurun_out/dense_dyn.txt 2>&1
cat gpurun_out/dense_dyn.txt
