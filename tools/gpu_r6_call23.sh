#!/bin/bash
# ordered HexPlane plane gradients: tests, then the field's backward and config #3 with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/hexord; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_deformation.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -8
for m in 0 1; do
  echo "GSR_HEX_ORDERED=$m"
  GSR_HEX_ORDERED=$m timeout 300 python tools/dev_hexviews.py 2>/dev/null | tail -1
  GSR_HEX_ORDERED=$m timeout 600 python tools/bench_config3.py 2>/dev/null | tail -1 | tee $O/config3_ord$m.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if 'ms_per_iteration' in k})"
done
timeout 900 python -m pytest tests/test_hip_configs.py -x -q -m gpu -p no:cacheprovider -k "config3" 2>&1 | grep -v Warning | tail -8
