#!/usr/bin/env python
"""render()-level timing of the fused prologue: gaussian_renderer.render + backward on a reference-shaped
GaussianModel (200k Gaussians, 640x480, 25 % dynamic with control-node deltas), prologue as torch kernels vs inside the HIP
kernels. Prints one JSON line; `python tools/bench_render_wrapper.py > profiles/r01_render_wrapper.json` on the GPU box."""
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
from test_hip_fused_prologue import _GaussianModel, _camera      # noqa: E402  (the reference-shaped model stand-in)

P, W, H = 200_000, 640, 480
cam = make_camera(W, H)
g = make_gaussians(P, cam, seed=0, sh_degree=0)
gc, gd = make_cotangents(cam, seed=1)
gc, gd = torch.tensor(gc, device="cuda"), torch.tensor(gd, device="cuda")
pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
out = {"workload": f"render() fwd+bwd, {P} Gaussians @{W}x{H}, 25% dynamic with dx/ds/dr, SH degree 0"}
for fused in (False, True):
    m = _GaussianModel(g, False, 0.25, seed=2)
    view = _camera(cam)
    K = int(m.dygs.sum())
    rng = np.random.default_rng(3)
    deltas = {k: torch.tensor(rng.normal(scale=s, size=(K, n)).astype(np.float32), device="cuda", requires_grad=True)
              for k, n, s in (("dx", 3, 0.002), ("ds", 3, 0.0001), ("dr", 4, 0.01))}
    leaves = list(m.leaves.values()) + list(deltas.values()) + [view.cam_rot_delta, view.cam_trans_delta]
    gr.FUSED_PROLOGUE = fused

    def step():
        for t in leaves:
            t.grad = None
        res = gr.render(view, m, pipe, bg, **deltas)
        torch.autograd.backward([res["render"], res["depth"]], [gc, gd])

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out["fused_prologue" if fused else "torch_prologue"] = {"ms_per_render_fwd_bwd": dt * 1e3, "gaussians_per_s": P / dt}
gr.FUSED_PROLOGUE = True
out["speedup"] = out["torch_prologue"]["ms_per_render_fwd_bwd"] / out["fused_prologue"]["ms_per_render_fwd_bwd"]
print(json.dumps(out))
