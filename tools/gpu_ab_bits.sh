#!/bin/bash
# gradient digests of every variant library + per-kernel timing of the variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for lib in 4dgs-slam_amd/_variants/*.so; do echo "== $lib"; GSR_GLUE=ctypes GSR_LIB=$PWD/$lib timeout 300 python tools/dev_grad_digest.py 2>&1 | tail -4; done
bash tools/dev_ab.sh
