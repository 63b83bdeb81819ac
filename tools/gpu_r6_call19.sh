#!/bin/bash
cd /root/repo
timeout 1700 python -m pytest tests/test_hip_deformation.py tests/test_hip_configs.py tests/test_hip_slam.py tests/test_hip_parity.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
python tools/bench_config3.py --modes batched --iters 3 2>/dev/null | tail -1
