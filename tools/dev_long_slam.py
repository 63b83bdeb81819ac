import json, os, sys, torch
REPO = "/root/repo"
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config
out = {}
for name, dyn, frames, wh in (("static_150", False, 150, (640, 480)), ("dynamic_100", True, 100, (320, 240))):
    torch.manual_seed(0)
    ds = SyntheticRGBDDataset(num_frames=frames, width=wh[0], height=wh[1], seed=0, dynamic=dyn, dystart=6 if dyn else None, spacing=0.02)
    cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                       "static_map_iters": 30, "dynamic_map_iters": 80, "network_init_iters": 50, "gaussian_update_every": 60,
                                                       "gaussian_update_offset": 20, "tracking_graph": True},
                                          "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                          "model_params": {"dynamic_model": dyn}})
    for i in range(len(ds)):
        ds[i]
    slam = SLAM(cfg, ds)
    res = slam.run()
    res["graph_stats"] = slam.frontend.graph_stats
    res["mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
    out[name] = {k: v for k, v in res.items() if k not in ("keyframes",)}
    out[name]["n_keyframes"] = len(res["keyframes"])
print(json.dumps(out, indent=1))
