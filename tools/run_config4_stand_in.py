#!/usr/bin/env python
"""BASELINE config #4's asset-free stand-in at the REFERENCE'S schedule (the run tests/test_hip_slam.py gates): 640x480, 40 frames with a
moving object from frame 6, configs/rgbd/tum/base_config.yaml value by value, the tracking graph, colour refinement, eval_rendering.
Prints one JSON document (-> profiles/rNN_config4_stand_in.json)."""
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset  # noqa: E402
from slam.system import SLAM, default_config, merge_config  # noqa: E402

torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=40, width=640, height=480, seed=0, dynamic=True, dystart=6, spacing=0.025)
cfg = merge_config(default_config(), {"Training": {"tracking_graph": True}, "model_params": {"dynamic_model": True}})
for i in range(len(ds)):
    ds[i]
with tempfile.TemporaryDirectory() as tmp:
    slam = SLAM(cfg, ds, save_dir=tmp)
    t0 = time.perf_counter()
    res = slam.run(color_refinement_iters=200)
    torch.cuda.synchronize()
    res["seconds_with_refinement_and_evaluation"] = time.perf_counter() - t0
t = cfg["Training"]
res["schedule"] = {k: t[k] for k in ("init_itr_num", "tracking_itr_num", "mapping_itr_num", "window_size", "kf_interval")}
res["schedule"]["dynamic_map_iters"] = slam.backend.dynamic_map_iters
res["graph_stats"] = slam.frontend.graph_stats
res["mapping_graph_stats"] = {"static": dict(getattr(slam.backend, "graph_stats", {}) or {}), "dynamic": dict(getattr(slam.backend, "dynamic_graph_stats", {}) or {}),
                              "initialize_map": dict(getattr(slam.backend, "init_graph_stats", {}) or {}),
                              "initialize_network": dict(getattr(slam.backend, "network_init_graph_stats", {}) or {})}
res["dynamic_gaussians"] = int(slam.gaussians.dygs.sum())
res["nodes"] = int(slam.gaussians.deform.deform.node_num)
print(json.dumps(res, indent=1, default=lambda o: o if isinstance(o, (int, float, str)) else str(o)))
