#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_control_nodes.py tests/test_node_losses.py tests/test_hip_slam.py -x -q 2>&1 | grep -v Warning | tail -3
python -m pytest tests/test_hip_slam.py -x -q 2>&1 | grep -v Warning | tail -3
for c in 1 0; do
GSR_CSR_REGISTERS=$c python tools/mapping_iteration_launches.py --wh 640 480 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
print(d['graph']['ms_per_iteration_without_capture'], d['device_us_per_iteration'], d['launches_per_iteration'])
for k,v in list(d['device_us_per_iteration_by_kernel'].items())[:40]:
    if 'csr' in k or 'Memset' in k: print('%8.1f %s'%(v,k[:90]))
"
done
} > gpurun_out/scratch.txt 2>&1
cat gpurun_out/scratch.txt
