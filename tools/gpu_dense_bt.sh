#!/bin/bash
R=/root/repo; O=$R/gpurun_out/r05; mkdir -p $O
python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/mapping_iteration_launches_static.json 2> /dev/null
python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic.json 2> /dev/null
GSR_DENSE_TRUNK=0 GSR_NETWORK_ADAM=0 python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic_library_trunk.json 2> /dev/null
python - <<'P'
import json
for f in ('mapping_iteration_launches_dynamic.json','mapping_iteration_launches_dynamic_library_trunk.json','mapping_iteration_launches_static.json'):
    d=json.load(open('/root/repo/gpurun_out/r05/'+f)); print(f, d['graph'] and d['graph']['ms_per_iteration_without_capture'], d['device_us_per_iteration'], d['launches_per_iteration'])
P
