#!/bin/bash
cd /root/repo
for l in 4dgs-slam_amd/_variants_timing*.so; do echo "== $l"; GSR_GLUE=ctypes GSR_LIB=$PWD/$l python tools/dev_fwd_timing.py 2>&1 | grep -v amdgpu.ids | grep "per-wave\|cycles per pair\|geometry"; done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_exact_math.py tests/test_hip_fused_prologue.py tests/test_hip_bindings.py -x -q -m gpu 2>&1 | tail -3
bash tools/dev_ab.sh
