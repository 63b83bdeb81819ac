#!/bin/bash
# quad slot summation in geometry_bwd (RAW): tests, tracking, SLAM demo, config #3 / #5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_fused_prologue.py tests/test_hip_exact_math.py tests/test_hip_views.py tests/test_hip_parity.py tests/test_hip_slam.py tests/test_hip_configs.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
python tools/bench_tracking.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print([(r['gaussians'], round(r['graph_us_per_iteration'],1)) for r in d['rows']])"
python tools/run_slam_demo.py --only graph 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['fps'],1), round(v['ate_rmse']*1000,2), round(v['before_opt']['mean_psnr'],2)) for k,v in d.items()})"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['m16']['ms_per_step'], d['config5']['ms_per_step'], d['config3']['ms_per_step'])"
bash tools/gpu_r6_call30.sh 2>&1 | grep "device busy\|geometry_bwd"
