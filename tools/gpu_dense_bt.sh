#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_control_nodes.py tests/test_hip_dense.py tests/test_node_losses.py tests/test_slam_losses.py -x -q 2>&1 | grep -v Warning | tail -12
} > gpurun_out/dense_dyn.txt 2>&1
cat gpurun_out/dense_dyn.txt
