#!/bin/bash
# round 4, GPU call 1: the mapping-iteration graph (tests + wall-time probes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_slam.py -x -q -k "hip_graph or outgrows or camera_step or slam_static or two_ranks_on_one_gpu" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -25 $O/tests.log
timeout 400 python tools/bench_backend_map.py > $O/backend_map_graph.json 2> $O/backend_map_graph.err; tail -3 $O/backend_map_graph.err; cat $O/backend_map_graph.json
timeout 400 python tools/bench_backend_map.py --eager > $O/backend_map_eager.json 2> $O/backend_map_eager.err; cat $O/backend_map_eager.json
timeout 500 python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/launches_static.json 2> $O/launches_static.err; tail -3 $O/launches_static.err; head -30 $O/launches_static.json
