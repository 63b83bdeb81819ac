"""dev: is a short DYNAMIC SLAM run bit-reproducible? Runs it twice in this process (fresh system each time) and compares the map and the node
network bitwise; prints the first differing tensor. Usage: python tools/dev_determinism.py [frames] [wh]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")]
import torch
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config


def run(frames, wh):
    torch.manual_seed(0)
    ds = SyntheticRGBDDataset(num_frames=frames, width=wh[0], height=wh[1], seed=1, dynamic=True, dystart=6)
    t = {"init_itr_num": 150, "init_gaussian_update": 100, "init_gaussian_reset": 120, "tracking_itr_num": 40, "static_map_iters": 20,
         "dynamic_map_iters": 30, "network_init_iters": 20, "gaussian_update_every": 60, "gaussian_update_offset": 20, "kf_interval": 4}
    cfg = merge_config(default_config(), {"Training": t, "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8},
                                          "opt_params": {"densify_from_iter": 100}, "model_params": {"dynamic_model": True}})
    slam = SLAM(cfg, ds)
    res = slam.run()
    g = slam.gaussians
    net = [p.detach().clone() for grp in g.deform.optimizer.param_groups for p in grp["params"]]
    return res, [p.detach().clone() for p in (g._xyz, g._features_dc, g._opacity, g._scaling, g._rotation)], net


frames = int(sys.argv[1]) if len(sys.argv) > 1 else 15
wh = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (160, 120)
a = run(frames, wh)
b = run(frames, wh)
print("run 1:", {k: a[0][k] for k in ("ate_rmse", "gaussians", "keyframes")}, a[0]["before_opt"]["mean_psnr"])
print("run 2:", {k: b[0][k] for k in ("ate_rmse", "gaussians", "keyframes")}, b[0]["before_opt"]["mean_psnr"])
same = True
for name, xs, ys in (("gaussians", a[1], b[1]), ("network", a[2], b[2])):
    for i, (x, y) in enumerate(zip(xs, ys)):
        if x.shape != y.shape or not torch.equal(x, y):
            same = False
            d = float((x - y).abs().max()) if x.shape == y.shape else None
            print(f"DIFFERENT {name}[{i}] shapes {tuple(x.shape)} {tuple(y.shape)} max |diff| {d}")
            break
print("BIT-IDENTICAL" if same else "NOT identical")
