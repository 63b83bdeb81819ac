#!/usr/bin/env python
"""BASELINE config #4 in its asset-free form: the whole SLAM system (slam/system.py) on the synthetic RGB-D sequence (slam/dataset.py),
static and dynamic, eager tracking and hipGraph tracking; prints ATE / PSNR / fps as one JSON document."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset  # noqa: E402
from slam.system import SLAM, default_config, merge_config  # noqa: E402

# --only <substring>: run the matching scenarios only; --profile: cProfile each run and print the 25 most expensive functions to stderr
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
profile = "--profile" in sys.argv
out = {}
for name, dyn, graph, frames, wh in (("static_640x480_eager", False, False, 40, (640, 480)), ("static_640x480_graph", False, True, 40, (640, 480)),
                                     ("dynamic_320x240_eager", True, False, 36, (320, 240)), ("dynamic_320x240_graph", True, True, 36, (320, 240)),
                                     ("dynamic_640x480_graph", True, True, 36, (640, 480))):
    if only not in name:
        continue
    torch.manual_seed(0)
    ds = SyntheticRGBDDataset(num_frames=frames, width=wh[0], height=wh[1], seed=0, dynamic=dyn, dystart=6 if dyn else None, spacing=0.025 if wh[0] > 320 else 0.03)
    cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                       "static_map_iters": 30, "dynamic_map_iters": 80, "network_init_iters": 50, "gaussian_update_every": 60,
                                                       "gaussian_update_offset": 20, "tracking_graph": graph,
                                                       "fused_grad_accumulation": os.environ.get("GSR_FUSED_ACC", "1") == "1",
                                                       "loss_values": os.environ.get("GSR_LOSS_VALUES", "0") == "1"},
                                          "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                          "model_params": {"dynamic_model": dyn}})
    for i in range(len(ds)):          # the sensor stream is "pre-recorded": rendering the synthetic frames is not part of the SLAM time
        ds[i]
    slam = SLAM(cfg, ds)
    if profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        res = pr.runcall(slam.run)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(20)
    else:
        res = slam.run()
    res["graph_stats"] = slam.frontend.graph_stats
    res["mapping_graph_stats"] = {"static": dict(getattr(slam.backend, "graph_stats", {}) or {}), "dynamic": dict(getattr(slam.backend, "dynamic_graph_stats", {}) or {}),
                              "initialize_map": dict(getattr(slam.backend, "init_graph_stats", {}) or {}),
                              "initialize_network": dict(getattr(slam.backend, "network_init_graph_stats", {}) or {})}
    res["resolution"] = list(wh)
    out[name] = res
print(json.dumps(out, indent=1))
