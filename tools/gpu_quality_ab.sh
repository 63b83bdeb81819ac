#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in 1 0; do GSR_FLOW_CLIPS=$c timeout 600 python tools/run_slam_demo.py --only dynamic_320x240_graph 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('clips=$c', k, round(v['seconds'],3), v['before_opt']['mean_psnr'], v['ate_rmse'], v['gaussians'], v['mapping_graph_stats']['dynamic'])
"; done
