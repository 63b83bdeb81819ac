#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
bash tools/dev_ab.sh
