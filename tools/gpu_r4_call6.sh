#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -v Warning $O/tests.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/c6/bench.json'))
print(d['ms_per_step'], d['kernel_us'])
PY
