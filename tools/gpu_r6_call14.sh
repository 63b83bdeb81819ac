#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
REPS=3 bash tools/dev_ab.sh > gpurun_out/r06/ab14.txt 2>&1
cat gpurun_out/r06/ab14.txt
GSR_BWD_PERSISTENT=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_bindings.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
