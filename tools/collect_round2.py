#!/usr/bin/env python
"""gpurun_out/<tag> (tools/profile_round2.sh / profile_round3.sh) + gpurun_out/counters_<tag> -> profiles/<tag>_* (tag = argv[1], default r02): rocprofv3 kernel stats, HBM traffic per launch
(FETCH_SIZE / WRITE_SIZE collected in separate passes, KiB units, gfx950 FETCH_SIZE half-count correction -- MI355X_MICROARCH.md 'HBM'),
the SQ counters of the tile kernels, and the JSON lines of the benches."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(REPO, "gpurun_out", tag), os.path.join(REPO, "profiles")
sys.path.insert(0, REPO)


def agg(pattern):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in d.items()}


stats_csv = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"].split("(")[0].replace("void ", "").split("<")[0]: r for r in csv.DictReader(open(stats_csv))}
fe, wr = agg(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv")), agg(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"))
sq = agg(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"))
out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                 "--no-secondary`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md); "
                 "fabric-side counters: Infinity-Cache hits are included", "kernels": {}}
for k in sorted(fe):
    if not k.startswith("gsr::"):
        continue
    short = k.split("::")[1].replace("_kernel", "")
    e = {"FETCH_SIZE_KiB": fe[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": wr.get(k, {}).get("WRITE_SIZE"),
         "hbm_bytes_per_launch": (2 * fe[k]["FETCH_SIZE"] + wr.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024,
         "rocprof_avg_us": float(stats[k]["AverageNs"]) / 1e3 if k in stats else None}
    e.update(sq.get(k, {}))
    out["kernels"][short] = e
out["render_bwd_bytes_per_launch"] = out["kernels"]["render_bwd"]["hbm_bytes_per_launch"]
# provenance: bench.py quotes this figure only when it runs on the kernel sources it was collected on (bench.csrc_digest)
import bench as bench_module
out["csrc_sha256"] = bench_module.csrc_digest()
out["head"] = subprocess.run(["git", "-C", REPO, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
bench = json.load(open(os.path.join(src, "bench.json")))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
out["render_bwd_traffic_over_algorithmic"] = out["render_bwd_bytes_per_launch"] / alg
out["whole_step_traffic_bytes"] = sum(v["hbm_bytes_per_launch"] for v in out["kernels"].values())
out["whole_step_traffic_over_algorithmic"] = out["whole_step_traffic_bytes"] / bench["roofline"]["whole_step_algorithmic_bytes"]
json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
for name in ("bench.json", "bench_under_rocprof.json", "long_lists.json", "tracking_graph.json", "config3.json", "render_wrapper.json", "slam_demo.json",
             "mapping_iteration.json", "mapping_iterationflow.json", "mapping_iterationnodes.json", "mapping_iterationnodesflow.json",
             "phase_cycles.json", "phase_cycles_slam_scale.json", "tracking_probe.json", "tracking_probe_autograd_route.json", "deform_mlp.jsonl", "views.json", "views_deltas.json", "views_100k.json", "backend_map.jsonl",
             "mapping_iteration_launches_dynamic.json", "mapping_iteration_launches_static.json", "dynamic_reproducibility.txt",
             "bench_two_ranks_one_gpu_gloo.json", "config4_stand_in.json", "mapping_iteration_launches_dynamic_library_trunk.json", "dense_layers.jsonl", "tile_timeline.json", "tile_timeline_tile_order.json",
             "dispatch_census.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name.replace('iterationflow', 'iteration_flow').replace('iterationnodesflow', 'iteration_nodes_flow').replace('iterationnodes', 'iteration_nodes')}"))
# bench.py quotes roofline.traffic from profiles/r02_hbm_traffic.json as committed when it ran; the copies kept here carry this run's figure
for name in ("bench.json", "bench_under_rocprof.json"):
    p = os.path.join(dst, f"{tag}_{name}")
    if os.path.exists(p):
        b = json.load(open(p))
        b["roofline"]["traffic"] = out["render_bwd_bytes_per_launch"]
        json.dump(b, open(p, "w"))
subprocess.run([sys.executable, os.path.join(REPO, "tools", "collect_counters.py"), tag, tag], check=True)
# provenance for bench.py's roofline.issue: the counters belong to the kernel sources they were collected on
cp = os.path.join(dst, f"{tag}_tile_kernel_counters.json")
if os.path.exists(cp):
    c = json.load(open(cp))
    c["csrc_sha256"], c["head"] = out["csrc_sha256"], out["head"]
    json.dump(c, open(cp, "w"), indent=1)
for name in ("phase_cycles.json", "phase_cycles_slam_scale.json"):
    pp = os.path.join(dst, f"{tag}_{name}")
    if os.path.exists(pp):
        c = json.load(open(pp))
        c["csrc_sha256"], c["head"] = out["csrc_sha256"], out["head"]
        json.dump(c, open(pp, "w"), indent=1)
# stdout of a multi-rank bench carries the ranks' transport chatter besides the line: keep the line
tp = os.path.join(dst, f"{tag}_bench_two_ranks_one_gpu_gloo.json")
if os.path.exists(tp):
    lines = [l for l in open(tp).read().splitlines() if l.startswith("{")]
    open(tp, "w").write((lines[-1] if lines else "") + "\n")
print(json.dumps({k: (round(v["rocprof_avg_us"], 1) if v["rocprof_avg_us"] else None, int(v["hbm_bytes_per_launch"])) for k, v in out["kernels"].items()}))
print("render_bwd traffic / algorithmic:", round(out["render_bwd_traffic_over_algorithmic"], 2), " whole step:", round(out["whole_step_traffic_over_algorithmic"], 2))
