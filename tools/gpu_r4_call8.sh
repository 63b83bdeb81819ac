#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/run_config4_stand_in.py > $O/config4.json 2> $O/config4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/c8/config4.json'))
print({k:d[k] for k in ('seconds','fps','gaussians','ate_rmse','before_opt','after_opt','seconds_with_refinement_and_evaluation')})
PY
timeout 900 python tools/run_slam_demo.py > $O/slam_demo.json 2> $O/slam_demo.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/c8/slam_demo.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:v[a] for a in v if a in ('fps','fps_after_init','ate_rmse','gaussians','seconds')}, v.get('before_opt',{}).get('mean_psnr'))
PY
timeout 600 python -m pytest tests/test_hip_slam.py -q -s -k "static_sequence_end_to_end or dynamic_sequence_end_to_end or config4_stand_in or tracking_graph_matches" 2>&1 | grep "^{" | head
