#!/bin/bash
# dW products of the node network: tests, library vs single vs many, block-count sweep, per-kernel split, and the dynamic iteration A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/wgm; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_dense.py -x -q -m gpu -p no:cacheprovider -k "wgrad" 2>&1 | grep -v Warning | tail -5
timeout 300 python tools/dev_wgrad_many.py 33280 66560 20800 2> $O/dev.err | tee $O/dev.jsonl
for b in 256 384 768 1024; do echo "blocks=$b"; GSR_WGM_BLOCKS=$b WGM_WHICH=many timeout 120 python tools/dev_wgrad_many.py 33280 2>/dev/null | tee -a $O/sweep.jsonl; done
( cd /tmp && WGM_WHICH=single,many rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o wgm -- python $GRAFT_REPO_ROOT/tools/dev_wgrad_many.py 33280 > /dev/null 2> $GRAFT_REPO_ROOT/$O/stats.err )
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/wgm/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "wgrad" in r["Name"]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
find $O -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' \) -delete
for m in 0 1; do
  GSR_DENSE_WGRAD_MANY=$m timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 > $O/launches_dynamic_many$m.json 2> $O/launches$m.err
  python - <<PY
import json
d = json.load(open("$O/launches_dynamic_many$m.json"))
print("many=$m", {k: d[k] for k in ("ms_per_iteration", "launches_per_iteration", "device_us_per_iteration")})
ks = d["device_us_per_iteration_by_kernel"]
print("  library GEMM us:", round(sum(v for k, v in ks.items() if k.startswith("Cijk")), 1), " wgrad us:", round(sum(v for k, v in ks.items() if "wgrad" in k), 1))
PY
done
timeout 600 python -m pytest tests/test_hip_dense.py tests/test_hip_slam.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -4
