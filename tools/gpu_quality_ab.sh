#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout 300 python tools/dev_quality_ab.py 36 320 240 2>/dev/null | grep QUALITY; }
TAG=default run
TAG=no_fused_reg GSR_FUSED_REGULARISERS=0 run
TAG=no_fused_trunk GSR_FUSED_TRUNK=0 run
TAG=neither GSR_FUSED_REGULARISERS=0 GSR_FUSED_TRUNK=0 run
TAG=old_layout TRAINING='{"dynamic_fixed_layout": false}' run
TAG=no_graphs TRAINING='{"mapping_graph": false}' run
TAG=dyn60 DYN_ITERS=60 run
TAG=dyn60_old_layout DYN_ITERS=60 TRAINING='{"dynamic_fixed_layout": false}' run
