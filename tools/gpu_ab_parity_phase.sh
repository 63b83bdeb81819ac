#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
bash tools/dev_ab.sh
GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py 2>/dev/null | grep "render_fwd"
