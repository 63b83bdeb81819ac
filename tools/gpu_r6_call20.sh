#!/bin/bash
# heads of the node network through the dense kernels: tests + the dynamic iteration's launch table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/heads; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_slam.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 > $O/launches_dynamic.json 2> $O/launches.err
python - <<PY
import json
d = json.load(open("$O/launches_dynamic.json"))
print({k: d[k] for k in ("ms_per_iteration", "launches_per_iteration", "device_us_per_iteration")})
ks = d["device_us_per_iteration_by_kernel"]
print("  library GEMM us:", round(sum(v for k, v in ks.items() if k.startswith("Cijk")), 1), " at::native us:", round(sum(v for k, v in ks.items() if "at::native" in k), 1), " rocprim us:", round(sum(v for k, v in ks.items() if "rocprim" in k), 1))
for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:45]: print("   %8.1f  %s" % (v, k[:110]))
PY
