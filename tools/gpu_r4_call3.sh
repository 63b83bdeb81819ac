#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_slam.py -x -q -k "outgrows" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -v Warning $O/tests.log | tail -8
timeout 400 python tools/bench_backend_map.py > $O/backend_map_graph.json 2> $O/backend_map_graph.err; cat $O/backend_map_graph.json
