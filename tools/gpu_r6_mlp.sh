#!/bin/bash
# dev: the fused deformation MLP alone -- fp32-MFMA kernel vs the bf16-split kernel (two row-tile settings), error against fp64 beside the time
cd /root/repo
for e in "GSR_MLP_FP32=1" "GSR_MLP_RT=4" "GSR_MLP_RT=2"; do echo "--- $e"; env $e python tools/dev_mlp_bench.py 2>&1 | tail -1; done
