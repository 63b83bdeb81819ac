#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_exact_math.py tests/test_hip_fused_prologue.py tests/test_hip_bindings.py -x -q -m gpu 2>&1 | tail -3
bash tools/dev_ab.sh
for l in 4dgs-slam_amd/_variants/*.so; do echo "== eager off $l"; GSR_EAGER_MAX=0 GSR_GLUE=ctypes GSR_LIB=$PWD/$l python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']; print(' '.join('%s %.1f' % (n[:8], v) for n, v in k.items()), 'sum %.1f' % sum(k.values()))"; done
