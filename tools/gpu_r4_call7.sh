#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_control_nodes.py tests/test_node_losses.py -x -q 2>&1 | grep -v Warning | tail -6
timeout 600 python tools/dev_determinism.py 2>&1 | grep -v Warning | tail -6
