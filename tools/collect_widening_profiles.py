#!/usr/bin/env python
"""Copies the outputs of tools/profile_widenings.sh from gpurun_out/ (scratch) into profiles/ (tracked): the rocprofv3 kernel
statistics of the deformation-network and control-node benches, the bench JSON lines, and the HBM traffic of the HexPlane
kernels (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, KiB, gfx950 FETCH_SIZE half-count correction --
/opt/skills/guides/MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"widen_{tag}")
dst = os.path.join(REPO, "profiles")


def one(pattern):
    return glob.glob(os.path.join(src, pattern), recursive=True)[0]


shutil.copy(one("def_stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_deformation_kernel_stats.csv"))
shutil.copy(one("nod_stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_control_nodes_kernel_stats.csv"))
bench = {}
for name in ("deformation_200k", "deformation_500k", "control_nodes_100k", "control_nodes_20k"):
    lines = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")]
    bench[name] = json.loads(lines[-1])


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in d.items()}


fe, wr = agg(one("def_fetch/**/*counter_collection.csv")), agg(one("def_write/**/*counter_collection.csv"))
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(one("def_stats/**/*kernel_stats.csv")))}
hbm = {}
for k in fe:
    if "gsr::" not in k:
        continue
    hbm[k] = {"FETCH_SIZE_KiB": fe[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": wr[k]["WRITE_SIZE"],
              "hbm_bytes_per_launch": (2 * fe[k]["FETCH_SIZE"] + wr[k]["WRITE_SIZE"]) * 1024,
              "rocprof_avg_us": float(stats[k]["AverageNs"]) / 1e3 if k in stats else None}
json.dump({"bench": bench, "hbm_traffic_n200k": hbm,
           "method": "tools/profile_widenings.sh; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch"},
          open(os.path.join(dst, f"{tag}_widenings_rank3.json"), "w"), indent=1)
for k, v in hbm.items():
    print(k, round(v["rocprof_avg_us"], 1), "us", round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB")
