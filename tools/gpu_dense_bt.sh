#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_fused_adam.py tests/test_capi_and_host.py -x -q -m gpu 2>&1 | grep -v Warning | tail -12
for a in 0 1; do echo "GSR_NETWORK_ADAM=$a"; GSR_NETWORK_ADAM=$a python tools/mapping_iteration_launches.py --dynamic 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
print(d['graph']['ms_per_iteration_without_capture'], d['device_us_per_iteration'], d['launches_per_iteration'], d['graph']['second_call'])
for k,v in list(d['device_us_per_iteration_by_kernel'].items())[:40]:
    if 'dam' in k or 'multi_tensor' in k: print('%8.1f %s'%(v,k))
"; done
} > gpurun_out/dense_dyn.txt 2>&1
cat gpurun_out/dense_dyn.txt
