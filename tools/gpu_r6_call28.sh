#!/bin/bash
# colour refinement, dynamic form: tests + the stand-in with refinement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_slam.py -x -q -m gpu -p no:cacheprovider -k "color_refinement or config4_stand_in" 2>&1 | grep -v Warning | tail -12
timeout 600 python tools/run_config4_stand_in.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('seconds','fps','seconds_with_refinement_and_evaluation','ate_rmse')}); print(d['before_opt']); print(d['after_opt'])"
