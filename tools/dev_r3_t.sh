#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_exact_math.py -x -q -m gpu 2>&1 | tail -3
for l in 4dgs-slam_amd/_variants_timing*.so; do echo "== $l"; GSR_GLUE=ctypes GSR_LIB=$PWD/$l python tools/dev_fwd_timing.py 2>&1 | grep -v amdgpu.ids | grep "per-wave" | head -1; done
bash tools/dev_ab.sh
