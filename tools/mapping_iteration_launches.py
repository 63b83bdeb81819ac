#!/usr/bin/env python
"""Dev probe: which source lines of a mapping iteration launch how many device kernels? Runs the SLAM demo for a few keyframes, then
`BackEnd.map(iters=N)` (or map_static with --static) under torch.profiler (with_stack) and attributes every kernel launch to the innermost
frame inside this repo. Usage: python tools/mapping_iteration_launches.py [--static] [--iters 6] [--frames 17] [--wh 320 240]"""
import collections
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset  # noqa: E402
from slam.system import SLAM, default_config, merge_config  # noqa: E402


def arg(name, default, n=1):
    if name not in sys.argv:
        return default
    i = sys.argv.index(name)
    vals = [int(v) for v in sys.argv[i + 1:i + 1 + n]]
    return vals[0] if n == 1 else vals


dyn = "--static" not in sys.argv
iters, frames, wh = arg("--iters", 6), arg("--frames", 17), arg("--wh", [320, 240], 2)
torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=frames, width=wh[0], height=wh[1], seed=0, dynamic=dyn, dystart=6 if dyn else None, spacing=0.03)
cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                   "static_map_iters": 30, "dynamic_map_iters": 80, "network_init_iters": 50, "gaussian_update_every": 60,
                                                   "gaussian_update_offset": 20, "tracking_graph": True},
                                      "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                      "model_params": {"dynamic_model": dyn}})
for i in range(len(ds)):
    ds[i]
slam = SLAM(cfg, ds)
slam.run()
be = slam.backend
window = list(be.current_window)
run = (lambda n: be.map(window, iters=n, dynamic_network=True)) if dyn else (lambda n: be.map_static(window, iters=n))
run(3)
torch.cuda.synchronize()
graph_wall = None
if "--eager" not in sys.argv:          # the plain iterations as hipGraph replays (slam/mapping_graph.py, slam/dynamic_graph.py): wall time of a long call
    n_long = 80 if dyn else 60
    stats = lambda: dict(getattr(be, "dynamic_graph_stats" if dyn else "graph_stats", {}) or {})
    t0 = time.perf_counter()
    run(n_long)
    torch.cuda.synchronize()
    graph_wall = {"iterations_per_call": n_long, "ms_per_iteration_incl_capture": (time.perf_counter() - t0) / n_long * 1e3}
    c0, s0 = stats().get("capture_ms", 0.0), stats()
    t0 = time.perf_counter()
    run(n_long)
    torch.cuda.synchronize()
    s1 = stats()
    graph_wall["ms_per_iteration_without_capture"] = ((time.perf_counter() - t0) * 1e3 - (s1.get("capture_ms", 0.0) - c0)) / n_long
    graph_wall["second_call"] = {k: s1[k] - s0.get(k, 0) for k in s1 if isinstance(s1[k], (int, float))}
    graph_wall["graph_stats"] = s1
    cfg["Training"]["mapping_graph"] = False           # the census below is of the directly executed iteration (what one replay replaces)
    be.config["Training"]["mapping_graph"] = False
t0 = time.perf_counter()
run(iters)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / iters
if "--cprofile" in sys.argv:          # host side only: where does the Python time of an iteration go?
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.runcall(run, iters * 3)
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    print("ms per iteration (unprofiled): %.3f" % (wall * 1e3))
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(45)
    raise SystemExit(0)
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes="--ops" in sys.argv or "--aten" in sys.argv,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    run(iters)
    torch.cuda.synchronize()
if "--aten" in sys.argv:             # every aten op with its input shapes: calls per iteration and device time (the small tensor-op kernels)
    rows = [(a.self_device_time_total / iters, a.count / iters, a.key, str(a.input_shapes)[:150]) for a in prof.key_averages(group_by_input_shape=True)
            if a.key.startswith("aten::") and a.self_device_time_total > 0]
    for us, cnt, key, shapes in sorted(rows, reverse=True)[:90]:
        print("%8.1f us %5.1f calls  %-28s %s" % (us, cnt, key, shapes), file=sys.stderr)
if "--ops" in sys.argv:              # which ops (with their input shapes) own the device time
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=90),
          file=sys.stderr)
by_line, by_op, ktime = collections.Counter(), collections.Counter(), collections.Counter()
by_region_n, by_region_us = collections.Counter(), collections.Counter()          # "gsr.<region>" ranges of slam/dynamic_graph.py; backward nodes by name
LAUNCH = ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipGraphLaunch", "hipModuleLaunchKernel", "hipExtLaunchKernel")
for e in prof.events():
    if e.device_type.name != "CPU":
        if not e.name.startswith(("gsr.", "Optimizer.step", "Memcpy", "Memset")) or e.name.startswith(("Memcpy", "Memset")):
            # (named ranges -- slam/dynamic_graph.py's regions, the optimizer's step -- show up on the device timeline too: they are not kernels)
            ktime[e.name[:70]] += getattr(e, "device_time", 0) or getattr(e, "cuda_time", 0)
        continue
    if not e.name.startswith(LAUNCH):
        continue
    p, op, where = e.cpu_parent, None, None          # walk up to the aten op / python frame that caused this launch
    top = e
    region = None
    while p is not None:
        if region is None and (p.name.startswith("gsr.") or "evaluate_function" in p.name):
            region = p.name.split("evaluate_function: ")[-1]
        if op is None and (p.name.startswith("aten::") or "Backward" in p.name):
            op = p.name
        if where is None:
            for fr in (p.stack or []):
                if "4dgs-slam_amd" in fr and "torch/" not in fr:
                    where = fr.split("4dgs-slam_amd/")[-1]
                    break
        top, p = p, p.cpu_parent
    if where is None:
        where = "(no python frame) " + top.name[:60]
    by_line[where] += 1
    by_op[(op or top.name)[:60]] += 1
    region = (region or "(outside) " + top.name)[:70]
    by_region_n[region] += 1
    by_region_us[region] += sum(getattr(k, "duration", 0) for k in (getattr(e, "kernels", None) or ()))
n_launch = sum(by_line.values())
out = {"graph": graph_wall, "dynamic": dyn, "resolution": wh, "gaussians": int(be.gaussians.get_xyz.shape[0]), "window": len(window), "ms_per_iteration": wall * 1e3,
       "launches_per_iteration": n_launch / iters,
       "by_source_line_per_iteration": {k: round(v / iters, 1) for k, v in by_line.most_common(50)},
       "by_op_per_iteration": {k: round(v / iters, 1) for k, v in by_op.most_common(40)},
       "by_region_per_iteration": {k: {"launches": round(by_region_n[k] / iters, 1), "device_us": round(v / iters, 1)} for k, v in by_region_us.most_common(60)},
       "device_us_per_iteration_by_kernel": {k: round(v / iters, 1) for k, v in ktime.most_common(30)},
       "device_us_per_iteration": round(sum(ktime.values()) / iters, 1)}
print(json.dumps(out, indent=1))
