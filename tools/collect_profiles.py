#!/usr/bin/env python
"""Copies the rocprofv3 outputs of tools/profile_round.sh from gpurun_out/ (scratch) into profiles/ (tracked) and derives
profiles/<tag>_hbm_traffic.json, which bench.py reads for roofline.traffic.

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected
in SEPARATE --pmc passes, both are in KiB, and on gfx950 FETCH_SIZE reports 1/2 of the fetched bytes, so
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   bytes per launch (averaged over the launches of the run).
Infinity-Cache hits are included in these fabric-side counters, so this is an upper bound on true HBM traffic."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"profile_{tag}")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in d.items()}


shutil.copy(os.path.join(src, "stats", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
for name in ("bench.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
fe = agg(os.path.join(src, "pmc_fetch", f"{tag}_counter_collection.csv"))
wr = agg(os.path.join(src, "pmc_write", f"{tag}_counter_collection.csv"))
sq = agg(os.path.join(src, "pmc_sq", f"{tag}_counter_collection.csv"))
stats = {r["Name"].split("(")[0].replace("void ", "").split("<")[0]: r for r in csv.DictReader(open(os.path.join(src, "stats", f"{tag}_kernel_stats.csv")))}
out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 20 --warmup 5`; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md)",
       "kernels": {}}
for k in sorted(fe):
    if not k.startswith("gsr::"):
        continue
    short = k.split("::")[1].replace("_kernel", "")
    e = {"FETCH_SIZE_KiB": fe[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": wr[k]["WRITE_SIZE"],
         "hbm_bytes_per_launch": (2 * fe[k]["FETCH_SIZE"] + wr[k]["WRITE_SIZE"]) * 1024,
         "rocprof_avg_us": float(stats[k]["AverageNs"]) / 1e3 if k in stats else None}
    e.update({c: v for c, v in sq.get(k, {}).items()})
    out["kernels"][short] = e
out["render_bwd_bytes_per_launch"] = out["kernels"]["render_bwd"]["hbm_bytes_per_launch"]
json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: (round(v["rocprof_avg_us"], 1), int(v["hbm_bytes_per_launch"])) for k, v in out["kernels"].items()}))
