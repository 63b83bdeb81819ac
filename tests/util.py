"""Shared helpers for the parity tests: run the same seeded scene through the oracle (CPU, checker) and through the
HIP product path (C ABI via the drop-in Python package) and compare."""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "4dgs-slam_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle  # noqa: E402  (test infrastructure)
from synthetic_scene import make_camera, make_gaussians, make_cotangents, keyframe_pose  # noqa: E402,F401


def rel_l1(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def oracle_run(g, cam, bg, gc=None, gd=None, colors_precomp=None, cov3D_precomp=None, dtype=np.float32, scale_modifier=1.0):
    kw = dict(bg=bg, means3D=g["means3D"], opacities=g["opacities"], viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
              campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=cam.H, image_width=cam.W,
              sh_degree=g.get("sh_degree", 0), scale_modifier=scale_modifier, dtype=dtype)
    if colors_precomp is not None:
        kw["colors_precomp"] = colors_precomp
    else:
        kw["shs"] = g["shs"]
    if cov3D_precomp is not None:
        kw["cov3D_precomp"] = cov3D_precomp
    else:
        kw["scales"] = g["scales"]
        kw["rotations"] = g["rotations"]
    out, st = oracle.rasterize_forward(**kw)
    grads = None
    if gc is not None:
        grads = oracle.rasterize_backward(st, projmatrix_raw=cam.projmatrix_raw, dL_dcolor=gc, dL_ddepth=gd,
                                          P=g["means3D"].shape[0], M=0 if colors_precomp is not None else g["shs"].shape[1],
                                          dtype=dtype)
    return out, st, grads


def hip_run(g, cam, bg, gc=None, gd=None, colors_precomp=None, cov3D_precomp=None, device="cuda", scale_modifier=1.0,
            pose_grad=True, return_state=False):
    """Through the drop-in API: GaussianRasterizer(...)(...) + autograd backward."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=device, requires_grad=rg)
    P = g["means3D"].shape[0]
    rs = GaussianRasterizationSettings(
        image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=T(bg), scale_modifier=scale_modifier,
        viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw),
        sh_degree=g.get("sh_degree", 0), campos=T(cam.campos), prefiltered=False, debug=False)
    need = gc is not None
    t = dict(means3D=T(g["means3D"], need), means2D=T(np.zeros((P, 3)), need), opacities=T(g["opacities"], need))
    if colors_precomp is not None:
        t["colors_precomp"] = T(colors_precomp, need)
    else:
        t["shs"] = T(g["shs"], need)
    if cov3D_precomp is not None:
        t["cov3D_precomp"] = T(cov3D_precomp, need)
    else:
        t["scales"] = T(g["scales"], need)
        t["rotations"] = T(g["rotations"], need)
    if pose_grad:
        t["theta"] = T(np.zeros(3), need)
        t["rho"] = T(np.zeros(3), need)
    color, radii, depth, opacity, n_touched = GaussianRasterizer(rs)(**t)
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(), opacity=opacity.detach().cpu().numpy(),
               radii=radii.cpu().numpy(), n_touched=n_touched.cpu().numpy())
    grads = None
    if need:
        loss = (color * T(gc)).sum() + (depth * T(gd)).sum()
        loss.backward()
        grads = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in t.items()}
    return out, grads


GRAD_KEYS = [  # (name in hip_run grads, name in oracle grads)
    ("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"),
    ("colors_precomp", "dL_dcolors"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"), ("cov3D_precomp", "dL_dcov3D"),
]


def compare(out_h, grads_h, out_o, grads_o, tag=""):
    """Returns a dict of metrics (BASELINE.md parity gate)."""
    m = dict(tag=tag)
    for k in ("color", "depth", "opacity"):
        m[k] = rel_l1(out_h[k], out_o[k])
    m["radii_mismatch"] = int((out_h["radii"] != out_o["radii"]).sum())
    m["visible_mismatch"] = int(((out_h["radii"] > 0) != (out_o["radii"] > 0)).sum())
    m["n_touched_mismatch"] = int((out_h["n_touched"] != out_o["n_touched"]).sum())
    m["n_touched_gt0_mismatch"] = int(((out_h["n_touched"] > 0) != (out_o["n_touched"] > 0)).sum())
    if grads_h is not None:
        for kh, ko in GRAD_KEYS:
            if grads_h.get(kh) is not None:
                m["g_" + kh] = rel_l1(grads_h[kh].reshape(-1), grads_o[ko].reshape(-1))
        tau = grads_o["dL_dtau"].sum(0)
        if grads_h.get("rho") is not None:
            m["g_rho"] = rel_l1(grads_h["rho"].reshape(-1), tau[:3])
            m["g_theta"] = rel_l1(grads_h["theta"].reshape(-1), tau[3:])
    return m
