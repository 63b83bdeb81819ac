#!/usr/bin/env python
"""Mapping-iteration-level timing of the two widenings of SURVEY.md 8f: gaussian_renderer.render + mapping loss + backward on a
reference-shaped GaussianModel (200k Gaussians, 640x480, 25 % dynamic with control-node deltas),
  "torch_*": prologue and loss as chains of torch kernels (what the reference does, restated), vs
  "fused_*": prologue inside the rasterizer kernels (rank 1) and the loss as two HIP kernels (rank 2).
Prints one JSON line; `python tools/bench_render_wrapper.py > profiles/r01_render_wrapper.json` on the GPU box."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import gaussian_renderer as gr                                   # noqa: E402
from synthetic_scene import make_camera, make_gaussians, make_cotangents   # noqa: E402
from synthetic_scene import GaussianModelStub as _GaussianModel, camera_namespace as _camera      # noqa: E402  (the reference-shaped model stand-in)
from slam_losses import get_loss_mapping, get_loss_tracking, mapping_loss_weights, tracking_loss_weights   # noqa: E402

P, W, H = 200_000, 640, 480
cam = make_camera(W, H)
g = make_gaussians(P, cam, seed=0, sh_degree=0)
gc, gd = make_cotangents(cam, seed=1)
gc, gd = torch.tensor(gc, device="cuda"), torch.tensor(gd, device="cuda")
pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
out = {"workload": f"render() fwd+bwd, {P} Gaussians @{W}x{H}, 25% dynamic with dx/ds/dr, SH degree 0"}
config = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
rng0 = np.random.default_rng(9)
gt_image = torch.tensor(rng0.uniform(0, 1, size=(3, H, W)).astype(np.float32), device="cuda")
gt_depth_np = rng0.uniform(0.3, 5.0, size=(H, W)).astype(np.float32)


def torch_mapping_loss(image, depth, vp):       # the reference's tensor expression (utils/slam_utils.py:252-364), masks recomputed per call
    gt_depth = torch.from_numpy(vp.depth).to(dtype=torch.float32, device=image.device)[None]
    w_rgb, w_dep = mapping_loss_weights(config, vp, vp.original_image, gt_depth)
    image_ab = torch.exp(vp.exposure_a) * image + vp.exposure_b
    return 0.9 * torch.abs(image_ab * w_rgb - vp.original_image * w_rgb).mean() + 0.1 * torch.abs(depth * w_dep - gt_depth * w_dep).mean()


for fused, with_loss in ((False, False), (True, False), (False, True), (True, True)):
    m = _GaussianModel(g, False, 0.25, seed=2)
    view = _camera(cam)
    view.original_image, view.depth, view.motion_mask = gt_image, gt_depth_np, None
    view.exposure_a = torch.nn.Parameter(torch.tensor([0.01], device="cuda"))
    view.exposure_b = torch.nn.Parameter(torch.tensor([0.0], device="cuda"))
    K = int(m.dygs.sum())
    rng = np.random.default_rng(3)
    deltas = {k: torch.tensor(rng.normal(scale=s, size=(K, n)).astype(np.float32), device="cuda", requires_grad=True)
              for k, n, s in (("dx", 3, 0.002), ("ds", 3, 0.0001), ("dr", 4, 0.01))}
    leaves = list(m.leaves.values()) + list(deltas.values()) + [view.cam_rot_delta, view.cam_trans_delta, view.exposure_a, view.exposure_b]
    gr.FUSED_PROLOGUE = fused

    def step():
        for t in leaves:
            t.grad = None
        res = gr.render(view, m, pipe, bg, **deltas)
        if not with_loss:
            torch.autograd.backward([res["render"], res["depth"]], [gc, gd])
        elif fused:
            get_loss_mapping(config, res["render"], res["depth"], view, res["opacity"]).backward()
        else:
            torch_mapping_loss(res["render"], res["depth"], view).backward()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    key = ("fused" if fused else "torch") + ("_prologue_and_loss" if with_loss else "_prologue")
    out[key] = {"ms_per_iteration": dt * 1e3, "gaussians_per_s": P / dt}

# ---- tracking iteration (utils/slam_frontend.py:405-440): render the static Gaussians (boolean mask), tracking loss, backward
def torch_tracking_loss(image, depth, opacity, vp):
    gt_depth = torch.from_numpy(vp.depth).to(dtype=torch.float32, device=image.device)[None]
    w_rgb, w_dep = tracking_loss_weights(config, vp, vp.original_image, gt_depth)
    image_ab = torch.exp(vp.exposure_a) * image + vp.exposure_b
    w_dep = w_dep * (opacity > 0.95)
    return 0.9 * (opacity * torch.abs(image_ab * w_rgb - vp.original_image * w_rgb)).mean() + 0.1 * torch.abs(depth * w_dep - gt_depth * w_dep).mean()


for fused in (False, True):
    m = _GaussianModel(g, False, 0.25, seed=2)
    view = _camera(cam)
    view.original_image, view.depth, view.motion_mask, view.uid = gt_image, gt_depth_np, None, 3
    view.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device="cuda")
    view.exposure_a = torch.nn.Parameter(torch.tensor([0.01], device="cuda"))
    view.exposure_b = torch.nn.Parameter(torch.tensor([0.0], device="cuda"))
    leaves = list(m.leaves.values()) + [view.cam_rot_delta, view.cam_trans_delta, view.exposure_a, view.exposure_b]
    static = m.dygs == False   # noqa: E712
    gr.FUSED_PROLOGUE = fused

    def track_step():
        for t in leaves:
            t.grad = None
        res = gr.render(view, m, pipe, bg, mask=static)
        loss = (get_loss_tracking if fused else lambda c, i, d, o, v: torch_tracking_loss(i, d, o, v))(config, res["render"], res["depth"], res["opacity"], view)
        loss.backward()

    for _ in range(10):
        track_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        track_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    out[("fused" if fused else "torch") + "_tracking_iteration"] = {"ms_per_iteration": dt * 1e3}
gr.FUSED_PROLOGUE = True
out["speedup_tracking_iteration"] = out["torch_tracking_iteration"]["ms_per_iteration"] / out["fused_tracking_iteration"]["ms_per_iteration"]
out["speedup_render"] = out["torch_prologue"]["ms_per_iteration"] / out["fused_prologue"]["ms_per_iteration"]
out["speedup_render_and_loss"] = out["torch_prologue_and_loss"]["ms_per_iteration"] / out["fused_prologue_and_loss"]["ms_per_iteration"]
print(json.dumps(out))
