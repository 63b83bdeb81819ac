#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_hip_deformation.py tests/test_hip_slam.py -x -q -m gpu -p no:cacheprovider -k "mlp or track or fused or network or latch" 2>&1 | grep -v Warning | tail -6
for e in "GSR_MLP_FP32=1" "GSR_MLP_RT=4" "GSR_MLP_RT=2"; do echo "--- $e"; env $e python tools/dev_mlp_bench.py 2>&1 | tail -1; done
echo "--- in_dim 64"; python tools/dev_mlp_bench.py 1000000 64 2>&1 | tail -1
