#!/usr/bin/env python
"""gpurun_out/counters_<tag>/ (tools/profile_counters.sh) -> profiles/<out>_tile_kernel_counters.json: per-launch averages of the SQ /
GRBM counters and rocprofv3's derived metrics for the rasterizer kernels, plus the ratios the roofline argument in DESIGN.md uses.

Units (MI355X_MICROARCH.md, 'rocprofv3 PMC slots' and the cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_BUSY_CYCLES is per shader engine (32 of them); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"counters_{tag}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gsr::" not in k:
            continue
        short = k.split("gsr::")[1].split("(")[0].split("<")[0].replace("_kernel", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if extra in r and r[extra] != "":
                acc[short]["_" + extra].append(float(r[extra]))
SIMDS, CUS = 1024, 256
out = {"method": "rocprofv3 --pmc <8 SQ counters per pass> --kernel-trace on `bench.py --steps 5 --warmup 2` (tools/profile_counters.sh); "
                 "per-launch averages. Derived rows: valu_issue_cycles_per_simd = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs; "
                 "kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_busy = valu_issue_cycles_per_simd / kernel_cycles (upper bound on how "
                 "busy the VALU is when one wave64 instruction holds a wave's issue port for a quad-cycle); valu_2cyc = 2 * SQ_INSTS_VALU / 1024 / "
                 "kernel_cycles (the SIMD-32 throughput view: 2 cycles of ALU per wave64 instruction); mean_waves_per_simd = 4 * SQ_WAVE_CYCLES / "
                 "(1024 * kernel_cycles)", "kernels": {}}
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    e = {n: c[n] for n in sorted(c)}
    if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d = {"kernel_cycles": cyc}
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_quadcycle_view"] = 4 * c["SQ_ACTIVE_INST_VALU"] / SIMDS / cyc
        if "SQ_INSTS_VALU" in c:
            d["valu_busy_2cycle_view"] = 2 * c["SQ_INSTS_VALU"] / SIMDS / cyc
            d["valu_instr_per_simd"] = c["SQ_INSTS_VALU"] / SIMDS
        if "SQ_WAVE_CYCLES" in c:
            d["mean_waves_per_simd"] = 4 * c["SQ_WAVE_CYCLES"] / (SIMDS * cyc)
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if n in c:
                    d["frac_of_wave_cycles_" + n] = c[n] / c["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            d["instr_per_wave"] = {n: c[n] / c["SQ_WAVES"] for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH",
                                                                       "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU_TRANS") if n in c}
        e["derived"] = d
    out["kernels"][k] = e
dst = os.path.join(REPO, "profiles", f"{out_tag}_tile_kernel_counters.json")
json.dump(out, open(dst, "w"), indent=1)
for k in ("render_fwd", "render_bwd"):
    if k in out["kernels"]:
        print(k, json.dumps(out["kernels"][k].get("derived", {}), indent=None)[:1500])
        print("  ", {n: round(v, 2) for n, v in out["kernels"][k].items() if not n.startswith("SQ_") and not n.startswith("GRBM") and n != "derived"})
