#!/usr/bin/env python
"""BASELINE config #3: P = 500 000 Gaussians, 8 keyframes, every Gaussian moved by the HexPlane deformation network
(HexPlane [64,64,64,25] x multires [1,2,4,8], width 64 -- arguments/__init__.py:76-94) through render(dynamic=True)
(gaussian_renderer/__init__.py:149-157), pose gradients on, mapping loss per view, one backward, Adam on the Gaussians and on the
network.  Reports the end-to-end iteration time and Gaussians x views / s for
  "reference_program": the reference's tensor program for the field (24 F.grid_sample per call) and nn.Linear layers, torch loss,
                       torch.optim.Adam, torch prologue -- around this repo's rasterizer;
  "fused":             HIP field + fused MLP, fused loss, FusedAdam -- one render(dynamic=True) per keyframe (rounds 1-4);
  "batched":           the keyframes of the iteration at once (round 5): render_views(dynamic=True) -- the deformation network evaluated once
                       for all times (spatial planes gathered once per Gaussian, one sort + one spatial scatter on the way back, the MLP over
                       all V * P rows), the multi-view rasterizer with the network's output as per-view deltas; fused Adam on the network.
Prints one JSON line (profiles/r05_config3.json).  usage: python tools/bench_config3.py [--P 500000] [--views 8] [--iters 5] [--modes fused,batched]"""
import argparse, json, os, sys, time, types
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import gaussian_renderer as gr
import deformation
import hexplane
from synthetic_scene import make_camera, make_gaussians, keyframe_pose
from synthetic_scene import GaussianModelStub as _GaussianModel, camera_namespace as _camera
hidden_params = deformation.default_hidden_params
from slam_losses import get_loss_mapping, mapping_loss_weights
from fused_adam import FusedAdam
from tools.bench_deformation import torch_field



def measure(modes, P=500_000, K=8, iters=5):
    """One JSON-able dict with the iteration time of every requested mode (see the module docstring)."""
    W, H = 640, 480
    config = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
    g = make_gaussians(P, make_camera(W, H), seed=0, sh_degree=0)
    rng = np.random.default_rng(11)
    out = {"workload": f"{K} keyframes x {P} Gaussians @{W}x{H}, render(dynamic=True) through the HexPlane deformation network, pose grads, "
                       "mapping loss, Adam (Gaussians + network)"}
    orig_features = hexplane.hexplane_features
    for mode in modes:
        fused, batched = mode != "reference_program", mode == "batched"
        torch.manual_seed(0)
        m = _GaussianModel(g, False, 0.0, seed=2)
        net = deformation.deform_network(hidden_params(bounds=8.0), "cuda").to("cuda")   # aabb that holds the synthetic scene (z up to 6)
        with torch.no_grad():
            for p_ in net.get_grid_parameters():
                if p_.requires_grad:
                    p_.mul_(0.05)                                    # small deltas: the scene stays renderable
        m._deformation = net
        views = []
        for k in range(K):
            R_w, t_w = keyframe_pose(k)
            v = _camera(make_camera(W, H, R=R_w, t=t_w))
            v.time = k / max(K - 1, 1) * 2 - 1
            v.original_image = torch.tensor(rng.uniform(0, 1, size=(3, H, W)).astype(np.float32), device="cuda")
            v.depth, v.motion_mask, v.uid = rng.uniform(0.3, 5.0, size=(H, W)).astype(np.float32), None, k
            v.exposure_a = torch.nn.Parameter(torch.tensor([0.0], device="cuda")); v.exposure_b = torch.nn.Parameter(torch.tensor([0.0], device="cuda"))
            views.append(v)
        groups = [{"params": [p_], "lr": lr, "name": n} for n, p_, lr in (("xyz", m._xyz, 1.6e-4), ("f_dc", m._features_dc, 2.5e-3),
                  ("opacity", m._opacity, 0.05), ("scaling", m._scaling, 1e-3), ("rotation", m._rotation, 1e-3))]
        opt = (FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
        net_params = [p_ for p_ in net.parameters() if p_.requires_grad]
        net_opt = torch.optim.Adam(net_params, lr=1.6e-4, eps=1e-15, fused=True) if batched else torch.optim.Adam(net_params, lr=1.6e-4, eps=1e-15)
        if fused:
            hexplane.hexplane_features = orig_features
        else:                                                        # the reference's field program + plain nn.Linear
            hexplane.hexplane_features = lambda pts, t, aabb, grids: torch_field(pts, t, aabb, [list(lv) for lv in grids])
            for mod in net.modules():
                if isinstance(mod, deformation.PointwiseLinear):
                    mod.forward = types.MethodType(lambda self, x: torch.nn.functional.linear(x, self.weight, self.bias), mod)

        def torch_loss(image, depth, vp):
            gt_depth = torch.from_numpy(vp.depth).to(dtype=torch.float32, device=image.device)[None]
            w_rgb, w_dep = mapping_loss_weights(config, vp, vp.original_image, gt_depth)
            image_ab = torch.exp(vp.exposure_a) * image + vp.exposure_b
            return 0.9 * torch.abs(image_ab * w_rgb - vp.original_image * w_rgb).mean() + 0.1 * torch.abs(depth * w_dep - gt_depth * w_dep).mean()

        def iteration():
            opt.zero_grad(set_to_none=True)
            net_opt.zero_grad(set_to_none=True)
            loss = 0.0
            if batched:
                for v, res in zip(views, gr.render_views(views, m, pipe, bg, dynamic=True)):
                    loss = loss + get_loss_mapping(config, res["render"], res["depth"], v, res["opacity"])
            else:
                for v in views:
                    res = gr.render(v, m, pipe, bg, dynamic=True)
                    loss = loss + (get_loss_mapping(config, res["render"], res["depth"], v, res["opacity"]) if fused else torch_loss(res["render"], res["depth"], v))
            loss.backward()
            opt.step()
            net_opt.step()

        for _ in range(2):
            iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            iteration()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        key = mode
        out[key + "_ms_per_iteration"] = round(dt * 1e3, 2)
        out[key + "_ms_per_view"] = round(dt * 1e3 / K, 3)
        out[key + "_gaussian_views_per_s"] = round(P * K / dt)
        assert all(torch.isfinite(p_.grad).all() for p_ in net_params if p_.grad is not None)
        del m, net, opt, net_opt, views
        torch.cuda.empty_cache()
    hexplane.hexplane_features = orig_features
    if "reference_program" in modes and "fused" in modes:
        out["speedup"] = round(out["reference_program_ms_per_iteration"] / out["fused_ms_per_iteration"], 2)
    if "batched" in modes and "fused" in modes:
        out["batched_over_fused"] = round(out["fused_ms_per_iteration"] / out["batched_ms_per_iteration"], 2)
    out["peak_memory_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=500_000)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--fused-only", action="store_true")
    ap.add_argument("--modes", default=None, help="comma list of reference_program, fused, batched (default: all three; --fused-only: fused, batched)")
    a = ap.parse_args()
    modes = a.modes.split(",") if a.modes else (["fused", "batched"] if a.fused_only else ["reference_program", "fused", "batched"])
    print(json.dumps(measure(modes, a.P, a.views, a.iters)))
