#!/bin/bash
# views_reduce / tau_sum load batching: tests + bench + dynamic iteration
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_views.py tests/test_hip_parity.py tests/test_hip_bindings.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -4
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-300
timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k: d[k] for k in ('ms_per_iteration','launches_per_iteration','device_us_per_iteration')}); ks=d['device_us_per_iteration_by_kernel']; print({k[:40]:v for k,v in ks.items() if 'views_reduce' in k or 'tau_sum' in k})"
timeout 600 python tools/bench_config3.py --modes batched 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if 'ms_per_iteration' in k})"
