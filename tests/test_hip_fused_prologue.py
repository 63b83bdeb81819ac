"""SURVEY.md 8f rank 1 -- the fused prologue: render() over a reference-shaped GaussianModel (raw parameters + activations) must
give the same image and the same parameter / delta / pose / screen-space gradients whether the activations, the delta scatter
and their chain rules run as torch kernels (the reference's way) or inside the HIP kernels (gsr_forward_raw / gsr_backward_raw)."""
import types

import numpy as np
import pytest
import torch

from util import make_camera, make_gaussians, make_cotangents, rel_l1

pytestmark = pytest.mark.gpu


from synthetic_scene import GaussianModelStub as _GaussianModel, camera_namespace as _camera   # noqa: E402


@pytest.mark.parametrize("isotropic,deg,with_deltas,with_mask", [(False, 0, False, False), (False, 2, True, False), (True, 1, True, False),
                                                                 (False, 3, False, False), (False, 1, False, True), (True, 0, True, True)])
def test_fused_prologue_matches_the_torch_prologue(isotropic, deg, with_deltas, with_mask):
    import gaussian_renderer as gr

    cam = make_camera(200, 152)
    g = make_gaussians(4000, cam, seed=31, sh_degree=deg)
    gc, gd = make_cotangents(cam, seed=32)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.3, 0.5], device="cuda")
    results = {}
    for fused in (False, True):
        m = _GaussianModel(g, isotropic, 0.3, seed=33)
        view = _camera(cam)
        K = int(m.dygs.sum())
        rng = np.random.default_rng(34)
        deltas = {}
        if with_deltas:
            deltas = {k: torch.tensor(rng.normal(scale=s, size=(K, n)).astype(np.float32), device="cuda", requires_grad=True)
                      for k, n, s in (("dx", 3, 0.02), ("ds", 3, 0.001), ("dr", 4, 0.05))}
        gr.FUSED_PROLOGUE = fused
        try:
            mask = (m.dygs == False) if with_mask else None   # noqa: E712  (the tracking call, slam_frontend.py:412-414)
            assert gr._fused_prologue_ok(m, pipe, mask, False) == fused
            res = gr.render(view, m, pipe, bg, mask=mask, **deltas)
        finally:
            gr.FUSED_PROLOGUE = True
        loss = (res["render"] * torch.tensor(gc, device="cuda")).sum() + (res["depth"] * torch.tensor(gd, device="cuda")).sum()
        loss.backward()
        grads = {k: v.grad for k, v in m.leaves.items()}
        grads.update({k: v.grad for k, v in deltas.items()})
        grads.update(theta=view.cam_rot_delta.grad, rho=view.cam_trans_delta.grad, viewspace=res["viewspace_points"].grad)
        results[fused] = (res, grads)
    (r0, g0), (r1, g1) = results[False], results[True]
    for k in ("render", "depth", "opacity"):
        assert rel_l1(r1[k].detach().cpu().numpy(), r0[k].detach().cpu().numpy()) < 1e-5, k
    assert torch.equal(r1["radii"], r0["radii"]) and (r1["n_touched"] != r0["n_touched"]).sum() <= 2
    for k in g0:
        if g0[k] is None or g0[k].numel() == 0:
            assert g1[k] is None or g1[k].numel() == 0 or float(g1[k].abs().sum()) == 0.0, k
            continue
        assert g1[k] is not None and g1[k].shape == g0[k].shape, k
        if float(g0[k].abs().max()) < 1e-9:          # analytically zero (rotation of an isotropic Gaussian): rounding noise on both sides
            assert float(g1[k].abs().max()) < 1e-9, k
            continue
        assert rel_l1(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 2e-4, (k, rel_l1(g1[k].cpu().numpy(), g0[k].cpu().numpy()))


def test_end_to_end_fit_reduces_the_loss():
    """The whole fused chain as a SLAM back-end would use it: render() from raw parameters -> fused mapping loss (+ fused SSIM) ->
    backward -> densification statistics -> FusedAdam, fitting a perturbed model to images rendered from the unperturbed one.
    Gradients that are right make the loss fall; this is the semantic check that goes beyond kernel-by-kernel parity."""
    import gaussian_renderer as gr
    from slam_losses import get_loss_mapping, ssim, add_densification_stats
    from fused_adam import FusedAdam

    torch.manual_seed(0)
    cam = make_camera(160, 120)
    g = make_gaussians(3000, cam, seed=7, sh_degree=0, scale_mean=0.02)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.0, 0.0, 0.0], device="cuda")
    config = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
    truth = _GaussianModel(g, False, 0.0, seed=8)
    views = []
    for k in range(3):
        from util import keyframe_pose
        R_w, t_w = keyframe_pose(k)
        v = _camera(make_camera(160, 120, R=R_w, t=t_w))
        with torch.no_grad():
            res = gr.render(v, truth, pipe, bg)
        v.original_image, v.depth, v.motion_mask, v.uid = res["render"].clone(), res["depth"][0].cpu().numpy(), None, k
        v.exposure_a = torch.nn.Parameter(torch.zeros(1, device="cuda")); v.exposure_b = torch.nn.Parameter(torch.zeros(1, device="cuda"))
        views.append(v)
    gp = dict(g)
    rng = np.random.default_rng(9)
    gp["means3D"] = g["means3D"] + rng.normal(scale=0.01, size=g["means3D"].shape).astype(np.float32)
    gp["shs"] = g["shs"] + rng.normal(scale=0.3, size=g["shs"].shape).astype(np.float32)
    m = _GaussianModel(gp, False, 0.0, seed=8)
    m.max_radii2D = torch.zeros(3000, device="cuda"); m.xyz_gradient_accum = torch.zeros(3000, 1, device="cuda"); m.denom = torch.zeros(3000, 1, device="cuda")
    opt = FusedAdam([{"params": [m._xyz], "lr": 2e-4, "name": "xyz"}, {"params": [m._features_dc], "lr": 1e-2, "name": "f_dc"},
                     {"params": [m._opacity], "lr": 2e-2, "name": "opacity"}, {"params": [m._scaling], "lr": 2e-3, "name": "scaling"},
                     {"params": [m._rotation], "lr": 1e-3, "name": "rotation"}], lr=0.0, eps=1e-15)
    history = []
    for it in range(60):
        opt.zero_grad(set_to_none=True)
        loss, pkgs = 0.0, []
        for v in views:
            res = gr.render(v, m, pipe, bg)
            loss = loss + 0.8 * get_loss_mapping(config, res["render"], res["depth"], v, res["opacity"]) + 0.2 * (1.0 - ssim(res["render"], v.original_image))
            pkgs.append(res)
        loss.backward()
        for res in pkgs:
            add_densification_stats(m, res["viewspace_points"], res["radii"])
        opt.step()
        history.append(float(loss.detach()))
    assert all(np.isfinite(history)) and history[-1] < 0.6 * history[0], (history[0], history[-1])
    assert float(m.denom.max()) == 60 * 3 and float(m.xyz_gradient_accum.sum()) > 0


class _RawModelFromGolden:
    """A reference-shaped model (raw parameters + activations) whose ACTIVATED values are the golden fixture's."""

    def __init__(self, fx, isotropic=False):
        L = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda", requires_grad=True)
        self._xyz = L(fx["means3D"])
        self._scaling = L(np.log(fx["scales"]))
        self._rotation = L(fx["rotations"])                                   # unit quaternions: normalize() is the identity on them
        op = fx["opacities"].astype(np.float64)
        self._opacity = L(np.log(op / (1 - op)))
        self._features_dc, self._features_rest = L(fx["shs"][:, :1]), L(fx["shs"][:, 1:])
        self.dygs = torch.tensor(fx["dygs"], device="cuda")
        self.active_sh_degree, self.max_sh_degree = int(fx["sh_degree"]), int(fx["max_sh_degree"])
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))


def test_fused_render_flow_reproduces_the_reference_golden(golden_dir):
    """render_flow through the fused raw route (projections, NDC flow, mask channel, scatter-adds and their chain
    rules inside the kernels) against golden_render_flow.npz, which was recorded from the reference's UNMODIFIED render_flow()."""
    import os
    import gaussian_renderer as gr
    from test_hip_wrapper_golden import _cam
    fx = np.load(os.path.join(golden_dir, "golden_render_flow.npz"))
    W, H = int(fx["W"]), int(fx["H"])
    cam, cam2 = _cam(fx["cam_k"], W, H, 3.0), _cam(fx["cam2_k"], W, H, 4.0)
    m = _RawModelFromGolden(fx)
    assert gr._flow_fused_ok(m)
    L = lambda k: torch.tensor(fx[k], device="cuda", requires_grad=True)
    ex = dict(dx=L("dx"), dx2=L("dx2"), ds=L("ds"), dr=L("dr"))
    calls = {"n": 0}
    real = gr._raw.rasterize_flow_raw
    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    gr._raw.rasterize_flow_raw = spy
    try:
        res = gr.render_flow(m, cam, cam2, ex["dx"], ex["dx2"], ex["dr"], ex["ds"])
    finally:
        gr._raw.rasterize_flow_raw = real
    assert calls["n"] == 1                                                  # the fused route was taken
    ((res["render"] * torch.tensor(fx["gc"], device="cuda")).sum() + (res["depth"] * torch.tensor(fx["gd"], device="cuda")).sum()).backward()
    for k in ("render", "depth", "alpha"):
        assert rel_l1(res[k].detach().cpu().numpy(), fx["out_" + k]) <= 1e-4, k
    assert (res["radii"].cpu().numpy() == fx["out_radii"]).all()
    assert (res["visibility_filter"].cpu().numpy() == fx["out_visibility_filter"]).all()
    assert rel_l1(m._xyz.grad.cpu().numpy(), fx["grad_xyz"]) <= 1e-3
    for name in ("_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
        assert getattr(m, name).grad is None, name                          # constants of render_flow (reference :307,326-334)
    assert rel_l1(res["viewspace_points"].grad.cpu().numpy(), fx["grad_viewspace"]) <= 1e-3
    for k, v in ex.items():
        assert rel_l1(v.grad.cpu().numpy(), fx["extra_" + k]) <= 1e-3, k


def test_fused_render_flow_matches_the_torch_chain_at_scale(isotropic=False):
    import gaussian_renderer as gr
    cam = make_camera(320, 240)
    from synthetic_scene import keyframe_pose
    R2, t2 = keyframe_pose(3)
    cam_b = make_camera(320, 240, R=R2, t=t2)
    g = make_gaussians(20000, cam, seed=41, sh_degree=0)
    gc, gd = make_cotangents(cam, seed=42)
    out = {}
    for fused in (False, True):
        m = _GaussianModel(g, isotropic, 0.25, seed=43)
        v1, v2 = _camera(cam), _camera(cam_b)
        K = int(m.dygs.sum())
        rng = np.random.default_rng(44)
        mk = lambda n, s: torch.tensor(rng.normal(scale=s, size=(K, n)).astype(np.float32), device="cuda", requires_grad=True)
        d1, d2, dr, ds = mk(3, 0.02), mk(3, 0.03), mk(4, 0.05), mk(3, 0.001)
        os_env = __import__("os").environ
        os_env["GSR_FUSED_FLOW"] = "1" if fused else "0"
        try:
            res = gr.render_flow(m, v1, v2, d1, d2, dr, ds)
        finally:
            os_env.pop("GSR_FUSED_FLOW", None)
        ((res["render"] * torch.tensor(gc, device="cuda")).sum() + (res["depth"] * torch.tensor(gd, device="cuda")).sum()).backward()
        out[fused] = (res, dict(xyz=m._xyz.grad, d1=d1.grad, d2=d2.grad, dr=dr.grad, ds=ds.grad, vs=res["viewspace_points"].grad))
    for k in ("render", "depth", "alpha"):
        assert rel_l1(out[True][0][k].detach().cpu().numpy(), out[False][0][k].detach().cpu().numpy()) <= 1e-5, k
    assert torch.equal(out[True][0]["radii"], out[False][0]["radii"])
    for k in out[True][1]:
        assert rel_l1(out[True][1][k].cpu().numpy(), out[False][1][k].cpu().numpy()) <= 3e-4, k


def test_fused_route_with_an_all_false_mask_returns_the_empty_render():
    """render(mask = all False) on the fused route must behave like the reference (x[mask] -> P = 0, rasterize_points.cu:85):
    zero image, empty radii / n_touched, zero gradients -- not an 'invalid argument' error."""
    import gaussian_renderer as gr
    cam = make_camera(96, 64)
    g = make_gaussians(500, cam, seed=3, sh_degree=0)
    m = _GaussianModel(g, False, 0.3, seed=4)
    view = _camera(cam)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    res = gr.render(view, m, pipe, torch.ones(3, device="cuda"), mask=torch.zeros(500, dtype=torch.bool, device="cuda"))
    assert res["radii"].numel() == 0 and res["n_touched"].numel() == 0
    assert float(res["render"].abs().max()) == 0.0 and float(res["depth"].abs().max()) == 0.0
    (res["render"].sum() + res["depth"].sum()).backward()
    assert float(m._xyz.grad.abs().sum()) == 0.0 and float(view.cam_rot_delta.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("deg,with_mask,binding", [(0, True, "native"), (3, False, "native"), (1, True, "ctypes")])
def test_pose_only_backward_gives_the_same_pose_and_screen_space_gradients(deg, with_mask, binding, monkeypatch):
    """GSR_BACKWARD_POSE_ONLY: rendering DETACHED Gaussians (camera tracking) makes the raw route's backward skip every parameter
    gradient. The pose gradient -- including the SH view-direction term -- and the screen-space gradient must be bit-identical to the
    full backward's."""
    import gaussian_renderer as gr
    from diff_gaussian_rasterization import _C
    if binding == "ctypes":
        monkeypatch.setattr(_C, "_glue", None)
    cam = make_camera(200, 152)
    g = make_gaussians(5000, cam, seed=41, sh_degree=deg)
    gc, gd = make_cotangents(cam, seed=42)
    bg = torch.tensor([0.1, 0.3, 0.5], device="cuda")
    out = {}
    for frozen in (False, True):
        m = _GaussianModel(g, False, 0.3, seed=43)
        view = _camera(cam)
        mask = (m.dygs == False) if with_mask else None   # noqa: E712
        pc = m
        if frozen:
            pc = types.SimpleNamespace(_xyz=m._xyz.detach(), _scaling=m._scaling.detach(), _rotation=m._rotation.detach(), _opacity=m._opacity.detach(),
                                       _features_dc=m._features_dc.detach(), _features_rest=m._features_rest.detach(), dygs=m.dygs,
                                       active_sh_degree=m.active_sh_degree)
        m2d = torch.zeros_like(m._xyz, requires_grad=True)
        image, radii, depth, opacity, n_touched = gr._render_fused(view, pc, bg, 1.0, m2d, None, None, None, mask, False)
        ((image * torch.tensor(gc, device="cuda")).sum() + (depth * torch.tensor(gd, device="cuda")).sum()).backward()
        out[frozen] = (view.cam_rot_delta.grad.clone(), view.cam_trans_delta.grad.clone(), m2d.grad.clone(), m._xyz.grad)
    assert out[True][3] is None and out[False][3] is not None
    assert float(out[False][0].abs().sum()) > 0 and float(out[False][1].abs().sum()) > 0
    for k in range(3):
        assert torch.equal(out[True][k], out[False][k]), k


@pytest.mark.parametrize("isotropic,deg,with_mask", [(False, 3, False), (True, 1, True), (False, 0, True)])
def test_fused_gradient_accumulation_on_the_raw_route_equals_autograd_bitwise(isotropic, deg, with_mask):
    """FusedAdam.enable_fused_gradient_accumulation(): three views, each back-propagated on its own, summed into the optimizer's flat
    gradient buffer by the backward kernels (GSR_BACKWARD_ACCUMULATE) -- against autograd's own accumulation of the returned gradients.
    Higher SH bands (read-modify-write at the store), the isotropic scale column and the mask gather (unselected rows untouched)."""
    import gaussian_renderer as gr
    from fused_adam import FusedAdam
    from util import keyframe_pose
    cam0 = make_camera(200, 152)
    g = make_gaussians(6000, cam0, seed=51, sh_degree=deg)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.3, 0.5], device="cuda")
    res = {}
    for fused in (True, False):
        m = _GaussianModel(g, isotropic, 0.3, seed=53)
        names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        tensors = (m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation)
        opt = FusedAdam([{"params": [p], "lr": 0.0, "name": n} for n, p in zip(names, tensors) if p.numel()], lr=0.0, eps=1e-15)
        opt.enable_fused_gradient_accumulation(fused)
        opt.zero_grad()
        mask = (m.dygs == False) if with_mask else None   # noqa: E712
        for k in (0, 2, 5):
            R, t = keyframe_pose(k)
            cam = make_camera(200, 152, R=R, t=t)
            gc, gd = make_cotangents(cam, seed=60 + k)
            out = gr.render(_camera(cam), m, pipe, bg, mask=mask)
            ((out["render"] * torch.tensor(gc, device="cuda")).sum() + (out["depth"] * torch.tensor(gd, device="cuda")).sum()).backward()
        res[fused] = [None if p.grad is None else p.grad.clone() for p in tensors]
        if fused:
            assert opt._bucket is not None and all(p.grad is v for p, v in zip(opt._bucket.params, opt._bucket.views))
    for a, b, n in zip(res[True], res[False], names):
        if b is None:
            assert a is None or float(a.abs().sum()) == 0.0, n
            continue
        assert a is not None and torch.equal(a, b), (n, float((a - b).abs().max()))
    assert float(res[True][0].abs().sum()) > 0
