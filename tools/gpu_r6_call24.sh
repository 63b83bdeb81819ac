#!/bin/bash
# tracking tail: tests + per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_slam.py -x -q -m gpu -k "track or latch or camera_step or camera" -p no:cacheprovider 2>&1 | grep -v Warning | tail -5
bash tools/gpu_r6_call22.sh 2>&1 | grep -v "at::native\|rocclr\|radix\|edge_int\|scan_kernel"
