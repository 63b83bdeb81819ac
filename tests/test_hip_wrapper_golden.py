"""GPU: this repo's render()/render_flow() over the HIP rasterizer must reproduce the golden vectors recorded from the
REFERENCE's unmodified wrapper running over the oracle (tests/golden/make_golden.py) -- the drop-in proof end to end."""
import os

import numpy as np
import pytest
import torch

from util import make_camera, keyframe_pose, rel_l1

pytestmark = pytest.mark.gpu


class _Cam:
    pass


def _cam(k, W, H, time):
    R, t = keyframe_pose(int(k))
    c = make_camera(W, H, R=R, t=t)
    cam = _Cam()
    T = lambda a: torch.tensor(a, device="cuda")
    cam.image_height, cam.image_width = H, W
    cam.FoVx, cam.FoVy = 2 * np.arctan(c.tanfovx), 2 * np.arctan(c.tanfovy)
    cam.world_view_transform, cam.full_proj_transform, cam.projection_matrix = T(c.viewmatrix), T(c.projmatrix), T(c.projmatrix_raw)
    cam.camera_center = T(c.campos)
    cam.cam_rot_delta = torch.nn.Parameter(torch.zeros(3, device="cuda"))
    cam.cam_trans_delta = torch.nn.Parameter(torch.zeros(3, device="cuda"))
    cam.time = time
    return cam


class _Model:
    def __init__(self, fx):
        L = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
        self._xyz, self._features, self._opac = L(fx["means3D"]), L(fx["shs"]), L(fx["opacities"])
        self._scal, self._rot = L(fx["scales"]), L(fx["rotations"])
        self.dygs = torch.tensor(fx["dygs"], device="cuda")
        self.active_sh_degree, self.max_sh_degree = int(fx["sh_degree"]), int(fx["max_sh_degree"])

    get_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: s._features)
    get_opacity = property(lambda s: s._opac)
    get_scaling = property(lambda s: s._scal)
    get_rotation = property(lambda s: s._rot)
    leaves = property(lambda s: dict(xyz=s._xyz, features=s._features, opacity=s._opac, scaling=s._scal, rotation=s._rot))


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False


@pytest.mark.parametrize("case", ["static", "tracking", "dynamic", "eval", "flow"])
def test_wrapper_over_hip_matches_reference_wrapper_over_oracle(case, golden_dir):
    import gaussian_renderer as gr
    import diff_gaussian_rasterization as dgr
    assert gr.GaussianRasterizer is dgr.GaussianRasterizer      # the product backend, not a stand-in
    fx = np.load(os.path.join(golden_dir, f"golden_render_{case}.npz"))
    W, H = int(fx["W"]), int(fx["H"])
    cam, cam2 = _cam(fx["cam_k"], W, H, 3.0), _cam(fx["cam2_k"], W, H, 4.0)
    m = _Model(fx)
    bg = torch.tensor(fx["bg"], device="cuda")
    pipe = _Pipe()
    L = lambda k: torch.tensor(fx[k], device="cuda", requires_grad=True)
    ex = {}
    if case == "static":
        res = gr.render(cam, m, pipe, bg, dx=0, ds=0, dr=None)
    elif case == "tracking":
        res = gr.render(cam, m, pipe, bg, dx=None, ds=None, dr=None, mask=(m.dygs == False))  # noqa: E712
    elif case == "dynamic":
        ex = dict(dx=L("dx"), ds=L("ds"), dr=L("dr"))
        res = gr.render(cam, m, pipe, bg, dx=ex["dx"], ds=ex["ds"], dr=ex["dr"])
    elif case == "eval":
        res = gr.render(cam, m, pipe, bg, dx=0, dr=0, ds=0)
    else:
        ex = dict(dx=L("dx"), dx2=L("dx2"), ds=L("ds"), dr=L("dr"))
        res = gr.render_flow(m, cam, cam2, ex["dx"], ex["dx2"], ex["dr"], ex["ds"])
    ((res["render"] * torch.tensor(fx["gc"], device="cuda")).sum() + (res["depth"] * torch.tensor(fx["gd"], device="cuda")).sum()).backward()
    for k in ("render", "depth") + (("opacity",) if case != "flow" else ("alpha",)):
        assert rel_l1(res[k].detach().cpu().numpy(), fx["out_" + k]) <= 1e-4, k
    assert (res["radii"].cpu().numpy() == fx["out_radii"]).all()
    assert (res["visibility_filter"].cpu().numpy() == fx["out_visibility_filter"]).all()
    if case != "flow":
        assert (res["n_touched"].cpu().numpy() != fx["out_n_touched"]).sum() <= 1
    for k, v in m.leaves.items():
        want = fx["grad_" + k]
        if want.size == 0:
            assert v.grad is None, k
        else:
            assert rel_l1(v.grad.cpu().numpy(), want) <= 1e-3, k
    assert rel_l1(res["viewspace_points"].grad.cpu().numpy(), fx["grad_viewspace"]) <= 1e-3
    for name, p in (("theta", cam.cam_rot_delta), ("rho", cam.cam_trans_delta)):
        want = fx["grad_" + name]
        if want.size:
            assert rel_l1(p.grad.cpu().numpy(), want) <= 1e-3, name
    for k, v in ex.items():
        assert rel_l1(v.grad.cpu().numpy(), fx["extra_" + k]) <= 1e-3, k
