#!/bin/bash
# gradient digests of every variant library + the parity suite on the product library + per-kernel timing of the variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for lib in 4dgs-slam_amd/_variants/*.so; do echo "== $lib"; GSR_GLUE=ctypes GSR_LIB=$PWD/$lib timeout 300 python tools/dev_grad_digest.py 2>&1 | tail -4; done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_views.py -x -q 2>&1 | tail -2
bash tools/dev_ab.sh
