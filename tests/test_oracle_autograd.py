"""The oracle's hand-derived backward (a restatement of DGR/cuda_rasterizer/backward.cu) against an INDEPENDENT
formulation: the pure-PyTorch forward of oracle/torch_raster.py differentiated by autograd in fp64."""
import numpy as np
import pytest
import torch

from util import oracle, oracle_run, make_camera, make_gaussians, make_cotangents, keyframe_pose, rel_l1
from oracle import torch_raster as tr

T64 = lambda a, rg=False: torch.tensor(np.asarray(a, np.float64), requires_grad=rg)


def _torch_run(g, cam, bg, gc, gd, colors_precomp=None, cov3D_precomp=None, view=None, proj=None, campos=None, scale_modifier=1.0):
    P = g["means3D"].shape[0]
    t = dict(means3D=T64(g["means3D"], True), means2D=T64(np.zeros((P, 3)), True), opacities=T64(g["opacities"], True))
    kw = {}
    if colors_precomp is not None:
        t["colors_precomp"] = T64(colors_precomp, True); kw["colors_precomp"] = t["colors_precomp"]
    else:
        t["shs"] = T64(g["shs"], True); kw["shs"] = t["shs"]
    if cov3D_precomp is not None:
        t["cov3D_precomp"] = T64(cov3D_precomp, True); kw["cov3D_precomp"] = t["cov3D_precomp"]
    else:
        t["scales"] = T64(g["scales"], True); t["rotations"] = T64(g["rotations"], True)
        kw["scales"], kw["rotations"] = t["scales"], t["rotations"]
    c, r, d, o, nt = tr.rasterize(t["means3D"], t["means2D"], t["opacities"], bg=T64(bg),
                                  viewmatrix=T64(cam.viewmatrix) if view is None else view,
                                  projmatrix=T64(cam.projmatrix) if proj is None else proj,
                                  campos=T64(cam.campos) if campos is None else campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                  image_height=cam.H, image_width=cam.W, sh_degree=g.get("sh_degree", 0), scale_modifier=scale_modifier, **kw)
    ((c * T64(gc)).sum() + (d * T64(gd)).sum()).backward()
    return dict(color=c.detach().numpy(), depth=d.detach().numpy(), opacity=o.detach().numpy(), radii=r.numpy(), n_touched=nt.numpy()), t


PAIRS = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"),
         ("colors_precomp", "dL_dcolors"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"), ("cov3D_precomp", "dL_dcov3D")]


@pytest.mark.parametrize("deg,W,H,mode", [(0, 96, 64, "sh"), (3, 100, 70, "sh"), (1, 70, 50, "precomp_color"), (2, 64, 48, "precomp_cov")])
def test_forward_and_gradients_match_autograd(deg, W, H, mode):
    cam = make_camera(W, H)
    g = make_gaussians(500, cam, seed=3, sh_degree=deg, scale_mean=0.02)
    gc, gd = make_cotangents(cam)
    bg = np.array([1.0, 0.5, 0.2])
    cp = np.random.default_rng(5).uniform(-1, 1, (500, 3)) if mode == "precomp_color" else None
    cov = None
    if mode == "precomp_cov":
        _, st0, _ = oracle_run(g, cam, bg, dtype=np.float64)
        cov = st0.state()["cov3D"]
    oo, st, go = oracle_run(g, cam, bg, gc, gd, colors_precomp=cp, cov3D_precomp=cov, dtype=np.float64, scale_modifier=1.3)
    ot, t = _torch_run(g, cam, bg, gc, gd, colors_precomp=cp, cov3D_precomp=cov, scale_modifier=1.3)
    for k in ("color", "depth", "opacity"):
        assert rel_l1(ot[k], oo[k]) < 1e-12, k
    assert (ot["radii"] == oo["radii"]).all() and (ot["n_touched"] == oo["n_touched"]).all()
    for kt, ko in PAIRS:
        if kt in t:
            ref = go[ko].reshape(-1)
            if kt == "scales":
                # reference quirk: dL/dscale is returned WITHOUT the scale_modifier factor (backward.cu:394-397 uses Rt and
                # dL_dMt before `s = mod * scale` is applied to them) -- the true gradient is mod x that.
                ref = ref * 1.3
            # scales/rotations/means carry the reference's 1/(det^2+1e-7) approximation (backward.cu:210): <= 1e-5
            assert rel_l1(ref, t[kt].grad.numpy().reshape(-1)) < 2e-5, kt


def test_pose_gradient_matches_autograd_through_se3_exp():
    """grad_rho/grad_theta vs d/dtau of the loss through W2C(tau) = SE3_exp(tau) @ W2C (utils/pose_utils.py:80-97), under the
    conditions where the reference's approximate pose Jacobians are exact: centred principal point, SH degree 0,
    no fov clamp. With the TUM principal point the documented approximation shows up as a ~0.4 % difference."""
    W, H, P = 96, 64, 600
    R, tt = keyframe_pose(3)
    res = {}
    for centred in (True, False):
        cam = make_camera(W, H, cx=W / 2 if centred else None, cy=H / 2 if centred else None, R=R, t=tt)
        g = make_gaussians(P, cam, seed=0, sh_degree=0, scale_mean=0.02)
        g["means3D"] = (g["means3D"].astype(np.float64) - tt) @ R
        gc, gd = make_cotangents(cam)
        bg = np.array([1.0, 0.5, 0.2])
        oo, st, go = oracle_run(g, cam, bg, gc, gd, dtype=np.float64)
        tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
        view = (tr.se3_exp(tau) @ T64(cam.viewmatrix).t()).t()
        proj = view @ T64(cam.projmatrix_raw)
        campos = torch.linalg.inv(view)[3, :3]
        _torch_run(g, cam, bg, gc, gd, view=view, proj=proj, campos=campos)
        res[centred] = rel_l1(go["dL_dtau"].sum(0), tau.grad.numpy())
    assert res[True] < 1e-5, res
    assert 1e-4 < res[False] < 3e-2, res   # the approximation is real and small


def test_fp32_oracle_close_to_fp64_oracle():
    cam = make_camera(160, 120)
    g = make_gaussians(3000, cam, seed=1, sh_degree=2)
    gc, gd = make_cotangents(cam)
    bg = np.ones(3)
    o32, _, g32 = oracle_run(g, cam, bg, gc, gd, dtype=np.float32)
    o64, _, g64 = oracle_run(g, cam, bg, gc, gd, dtype=np.float64)
    for k in ("color", "depth", "opacity"):
        assert rel_l1(o32[k], o64[k]) < 1e-5
    for k in g32:
        if g64[k].size and np.abs(g64[k]).sum() > 0:
            assert rel_l1(g32[k], g64[k]) < 1e-4, k
