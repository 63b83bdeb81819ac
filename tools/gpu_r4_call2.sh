#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_slam.py -x -q -k "hip_graph or outgrows" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -v Warning $O/tests.log | tail -25
timeout 400 python tools/bench_backend_map.py > $O/backend_map_graph.json 2> $O/backend_map_graph.err; cat $O/backend_map_graph.json
timeout 500 python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/launches_static.json 2> $O/launches_static.err; head -16 $O/launches_static.json
