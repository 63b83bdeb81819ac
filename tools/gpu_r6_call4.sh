#!/bin/bash
# round 6, GPU call 4: production item ordering: A/B (GSR_ORDER_ITEMS=0/1 on one library), parity + multi-view tests
cd /root/repo; mkdir -p gpurun_out/r06
REPS=3 bash tools/dev_ab.sh > gpurun_out/r06/ab4.txt 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_views.py tests/test_hip_bindings.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06/tests4.txt 2>&1
cat gpurun_out/r06/ab4.txt; tail -5 gpurun_out/r06/tests4.txt
