#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_configs.py -x -q -k "bench or rccl" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -v Warning $O/tests.log | tail -30
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err; cat $O/bench.json
