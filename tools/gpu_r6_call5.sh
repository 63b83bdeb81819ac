#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
L=$PWD/4dgs-slam_amd/_variants
for v in tl_work tl_monly; do
  for o in 0 1; do
    GSR_ORDER_ITEMS=$o GSR_GLUE=ctypes GSR_LIB=$L/$v.so python tools/tile_timeline.py --json > gpurun_out/r06/tl_${v}_$o.json 2>/dev/null
  done
done
python - <<'PY'
import json
for v in ("tl_work", "tl_monly"):
    for o in (0, 1):
        k = json.load(open(f"gpurun_out/r06/tl_{v}_{o}.json"))[1]
        print(v, "order", o, {x: k[x] for x in ("first_start_to_last_end_us", "block_us_mean_p50_p90_max", "last_block_starts_at_us", "tail_us", "mean_resident_blocks", "resident_blocks_by_decile")})
PY
