#!/bin/bash
# round 6: SH rows through LDS: tests + per-kernel times at SH degree 3 (M = 16), option on / off
cd /root/repo; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_sh_rows.py tests/test_hip_parity.py tests/test_hip_fused_prologue.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
for rep in 1 2; do
  for lib in rows0 rows1; do
    for deg in 3 2; do
      env $(cat 4dgs-slam_amd/_variants/$lib.env 2>/dev/null) GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_variants/$lib.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --sh-degree $deg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']; print('$lib deg $deg', ' '.join('%s %.1f' % (n[:8], v) for n, v in k.items()), 'sum %.1f' % sum(k.values()))
"
    done
  done
done
