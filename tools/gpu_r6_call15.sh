#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_views.py tests/test_hip_bindings.py tests/test_hip_fused_prologue.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
bash tools/dev_traffic.sh 2>&1 | tail -12
REPS=2 bash tools/dev_ab.sh 2>&1 | tail -6
