"""dev: where does the dynamic branch lose quality? PSNR on static vs moving pixels, keyframes vs other frames."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config
from gaussian_renderer import render
torch.manual_seed(0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 240)
ds = SyntheticRGBDDataset(num_frames=36, width=W, height=H, seed=0, dynamic=True, dystart=6, spacing=0.03)
cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                   "static_map_iters": 30, "dynamic_map_iters": int(os.environ.get("DYN_ITERS", 80)), "network_init_iters": 50, "gaussian_update_every": 60,
                                                   "gaussian_update_offset": 20},
                                      "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                      "model_params": {"dynamic_model": True}})
slam = SLAM(cfg, ds)
res = slam.run()
print({k: res[k] for k in ("fps", "ate_rmse", "gaussians", "keyframes")}, res["before_opt"])
g, be = slam.gaussians, slam.backend
rows = []
for idx, cam in slam.frontend.cameras.items():
    gt, gtd, _, mm = ds[idx]
    with torch.no_grad():
        dx, ds_, dr = be._deltas(cam, train=False) if g.deform_init else (None, None, None)
        pkg = render(cam, g, slam.pipeline_params, slam.background, dx=dx, ds=ds_, dr=dr)
    img = pkg["render"].clamp(0, 1); gt = gt.to(img.device)
    err = ((img - gt) ** 2).mean(0)
    static = torch.ones_like(err, dtype=torch.bool) if mm is None else torch.as_tensor(mm, device=img.device).bool()
    ps = lambda m: float(-10 * torch.log10(err[m].mean())) if m.any() else float("nan")
    rows.append((idx, idx in slam.frontend.kf_indices, ps(static), ps(~static), float((~static).float().mean()), int((pkg["radii"] > 0).sum())))
for r in rows:
    print("frame %2d kf=%d  psnr static px %.1f  moving px %.1f  (moving share %.3f)  visible %d" % r)
print("dygs", int(g.dygs.sum()), "of", g.get_xyz.shape[0], "nodes", g.deform.deform.node_num if g.deform is not None else 0)
