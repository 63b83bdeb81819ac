#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
REPS=2 bash tools/dev_ab.sh > gpurun_out/r06/ab11.txt 2>&1
GSR_DEAL_HEAVY=1 GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_variants/timeline.so python tools/tile_timeline.py --json --raw gpurun_out/r06/spans_raw3.npz > gpurun_out/r06/timeline_h1.json 2>/dev/null
cat gpurun_out/r06/ab11.txt
python -c "
import json
for k in json.load(open('gpurun_out/r06/timeline_h1.json')): print({x: k[x] for x in ('kernel', 'first_start_to_last_end_us', 'block_us_mean_p50_p90_max', 'tail_us', 'mean_resident_blocks', 'resident_blocks_by_decile')})
"
