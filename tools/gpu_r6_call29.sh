#!/bin/bash
# begin_iteration through the fused trunk: SLAM tests, control-node tests, the stand-in with refinement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_hip_slam.py tests/test_hip_control_nodes.py tests/test_hip_dense.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -8
timeout 600 python tools/run_config4_stand_in.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('seconds','fps','seconds_with_refinement_and_evaluation','ate_rmse')}); print(d['before_opt']); print(d['after_opt'])"
