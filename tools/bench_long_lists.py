#!/usr/bin/env python
"""SLAM-shaped workloads next to BASELINE config #2: same P and resolution, larger Gaussians (scale_mean 0.02 ... 0.05 ->
tile lists of thousands of entries, 17-72 instances per Gaussian) and SH degree 3. Runs bench.py per point and reports per-kernel us and
us per million instances.      gpurun -- 'python tools/bench_long_lists.py > gpurun_out/long_lists.json'"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for scale_mean, deg in ((0.005, 0), (0.005, 3), (0.02, 0), (0.02, 3), (0.03, 3), (0.05, 3)):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "long", "--scale-mean", str(scale_mean), "--sh-degree", str(deg),
                        "--steps", "30", "--warmup", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        rows.append({"scale_mean": scale_mean, "sh_degree": deg, "error": r.stderr[-300:]})
        continue
    d = json.loads(line[-1])
    R = d["config"]["instances"]
    rows.append({"scale_mean": scale_mean, "sh_degree": deg, "instances": R, "visible": d["config"]["visible"], "instances_per_gaussian": R / 200000,
                 "mean_tile_list": R / 1200, "ms_per_step": d["ms_per_step"], "ms_per_step_nonspeculative": d.get("ms_per_step_nonspeculative"),
                 "kernel_us": d["kernel_us"], "us_per_million_instances": {k: v / (R / 1e6) for k, v in d["kernel_us"].items()},
                 "step_ns_per_instance": d["ms_per_step"] * 1e6 / R})
print(json.dumps({"what": "200k Gaussians @640x480, fwd+bwd; rows = (scale_mean, SH degree); kernel_us from HIP events", "rows": rows}, indent=1))
