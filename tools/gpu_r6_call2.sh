#!/bin/bash
# round 6, GPU call 2: forward prefetch + longest-first work items: A/B, timelines, parity of the combination
cd /root/repo; mkdir -p gpurun_out/r06
REPS=3 bash tools/dev_ab.sh > gpurun_out/r06/ab2.txt 2>&1
L=$PWD/4dgs-slam_amd/_variants
GSR_GLUE=ctypes GSR_LIB=$L/timeline.so python tools/tile_timeline.py --json > gpurun_out/r06/timeline_pf.json 2> gpurun_out/r06/timeline.err
GSR_ORDER_ITEMS=1 GSR_GLUE=ctypes GSR_LIB=$L/timeline.so python tools/tile_timeline.py --json > gpurun_out/r06/timeline_pf_order.json 2>> gpurun_out/r06/timeline.err
GSR_ORDER_ITEMS=1 GSR_GLUE=ctypes GSR_LIB=$L/pf.so timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu > gpurun_out/r06/parity_pf_order.txt 2>&1
cat gpurun_out/r06/ab2.txt; python - <<'PY'
import json
for f in ("timeline_pf", "timeline_pf_order"):
    for k in json.load(open(f"gpurun_out/r06/{f}.json")):
        print(f, {x: k[x] for x in ("kernel", "first_start_to_last_end_us", "block_us_mean_p50_p90_max", "last_block_starts_at_us", "tail_us", "mean_resident_blocks", "resident_blocks_by_decile")})
PY
tail -5 gpurun_out/r06/parity_pf_order.txt
