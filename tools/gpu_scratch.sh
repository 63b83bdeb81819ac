#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_control_nodes.py tests/test_node_losses.py tests/test_hip_slam.py -x -q 2>&1 | grep -v Warning | tail -5
python tools/mapping_iteration_launches.py --wh 640 480 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
print(d['graph']['ms_per_iteration_without_capture'], d['device_us_per_iteration'], d['launches_per_iteration'], d['graph']['second_call'])
for k,v in list(d['device_us_per_iteration_by_kernel'].items())[:12]: print('%8.1f %s'%(v,k[:90]))
print({k:v['launches'] for k,v in d['by_region_per_iteration'].items() if k.startswith('gsr.')})
"
} > gpurun_out/scratch.txt 2>&1
cat gpurun_out/scratch.txt
