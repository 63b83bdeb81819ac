"""dev: a longer dynamic SLAM run (more keyframes than the window holds, random keyframes from a growing pool): graph statistics, fps, memory."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=frames, width=640, height=480, seed=0, dynamic=True, dystart=6, spacing=0.012)
cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                   "static_map_iters": 30, "dynamic_map_iters": 80, "network_init_iters": 50, "gaussian_update_every": 60,
                                                   "gaussian_update_offset": 20, "tracking_graph": True},
                                      "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                      "model_params": {"dynamic_model": True}})
for i in range(len(ds)):
    ds[i]
slam = SLAM(cfg, ds)
res = slam.run()
be = slam.backend
print(json.dumps({"frames": frames, "keyframes": len(res["keyframes"]), "fps": res["fps"], "ate_mm": res["ate_rmse"] * 1e3, "psnr": res["before_opt"]["mean_psnr"],
                  "gaussians": res["gaussians"], "dynamic": be.dynamic_graph_stats, "static": getattr(be, "graph_stats", None), "init": getattr(be, "init_graph_stats", None),
                  "tracking": slam.frontend.graph_stats, "reserved_GB": torch.cuda.memory_reserved() / 2**30, "allocated_GB": torch.cuda.memory_allocated() / 2**30}))
