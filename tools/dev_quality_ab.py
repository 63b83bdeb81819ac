"""dev: PSNR / ATE of the short dynamic run of tools/dev_determinism.py under the environment's toggles (one run)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config
frames, wh = int(sys.argv[1]), (int(sys.argv[2]), int(sys.argv[3]))
torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=frames, width=wh[0], height=wh[1], seed=1, dynamic=True, dystart=6)
t = {"init_itr_num": 150, "init_gaussian_update": 100, "init_gaussian_reset": 120, "tracking_itr_num": 40, "static_map_iters": 20,
     "dynamic_map_iters": int(os.environ.get("DYN_ITERS", 30)), "network_init_iters": 20, "gaussian_update_every": 60, "gaussian_update_offset": 20, "kf_interval": 4}
t.update(json.loads(os.environ.get("TRAINING", "{}")))
cfg = merge_config(default_config(), {"Training": t, "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8},
                                      "opt_params": {"densify_from_iter": 100}, "model_params": {"dynamic_model": True}})
res = SLAM(cfg, ds).run()
print("QUALITY", os.environ.get("TAG", ""), round(res["before_opt"]["mean_psnr"], 3), round(res["ate_rmse"] * 1e3, 2), "mm", res["gaussians"], res["keyframes"])
