#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 > gpurun_out/launches_dynamic.json 2> gpurun_out/launches_dynamic.err; tail -3 gpurun_out/launches_dynamic.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/launches_dynamic.json"))
print({k: d[k] for k in ("graph", "ms_per_iteration", "launches_per_iteration", "device_us_per_iteration", "gaussians", "window")})
PY
timeout 900 python tools/run_slam_demo.py --only dynamic > gpurun_out/slam_demo_dynamic.json 2> gpurun_out/slam_demo_dynamic.err; tail -3 gpurun_out/slam_demo_dynamic.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/slam_demo_dynamic.json"))
for k, v in d.items():
    print(k, {a: v[a] for a in ("seconds", "fps", "ate_rmse", "gaussians")}, v["before_opt"]["mean_psnr"])
PY
