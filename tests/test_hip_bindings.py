"""Both host bindings of the C ABI on the GPU (the driver's run only ever used the native glue): the same small scene
through GSR_GLUE=native and GSR_GLUE=ctypes in two processes -- plain rasterizer API, raw (fused prologue) route with deltas and a
mask, fused tracking loss, FusedAdam -- must give bitwise the same images and gradients (same kernels, only the marshalling differs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, json, os, sys, types
import numpy as np
sys.path[:0] = [os.path.join(r"{repo}", "tests"), r"{repo}", os.path.join(r"{repo}", "4dgs-slam_amd")]
import torch
from util import make_camera, make_gaussians, make_cotangents, hip_run
from diff_gaussian_rasterization import _C
import gaussian_renderer as gr, slam_losses
from fused_adam import FusedAdam
from synthetic_scene import GaussianModelStub, camera_namespace
h = hashlib.sha256()
def eat(t):
    if t is not None:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t).tobytes())
cam = make_camera(160, 120)
g = make_gaussians(4000, cam, seed=1, sh_degree=1)
gc, gd = make_cotangents(cam, seed=2)
out, grads = hip_run(g, cam, np.array([0.1, 0.2, 0.3], np.float32), gc, gd)
for k in sorted(out): eat(out[k])
for k in sorted(grads): eat(grads[k])
m = GaussianModelStub(g, False, 0.3, seed=3)
view = camera_namespace(cam)
K = int(m.dygs.sum()); rng = np.random.default_rng(4)
mk = lambda n, s: torch.tensor(rng.normal(scale=s, size=(K, n)).astype(np.float32), device="cuda", requires_grad=True)
dx, ds, dr = mk(3, 0.02), mk(3, 0.001), mk(4, 0.05)
pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
res = gr.render(view, m, pipe, torch.ones(3, device="cuda"), dx=dx, ds=ds, dr=dr)
view.uid, view.motion_mask = 1, None
view.original_image = torch.tensor(rng.uniform(0, 1, (3, 120, 160)).astype(np.float32), device="cuda")
view.depth = rng.uniform(0.5, 4, (120, 160)).astype(np.float32)
view.grad_mask = torch.tensor(rng.uniform(size=(1, 120, 160)) < 0.5, device="cuda")
view.exposure_a = torch.nn.Parameter(torch.zeros(1, device="cuda")); view.exposure_b = torch.nn.Parameter(torch.zeros(1, device="cuda"))
cfg = {{"Training": {{"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}}}
loss = slam_losses.get_loss_tracking(cfg, res["render"], res["depth"], res["opacity"], view, rm_dynamic=True)
loss = loss + 0.1 * (1 - slam_losses.ssim(res["render"], view.original_image))
loss.backward()
for t in (res["render"], res["depth"], loss, m._xyz.grad, m._scaling.grad, m._rotation.grad, m._opacity.grad, m._features_dc.grad, dx.grad, ds.grad, dr.grad,
          view.cam_rot_delta.grad, view.cam_trans_delta.grad, view.exposure_a.grad):
    eat(t)
opt = FusedAdam([{{"params": [p], "lr": 1e-3}} for p in (m._xyz, m._scaling, m._rotation, m._opacity, m._features_dc)], lr=0.0, eps=1e-15)
opt.step()
eat(m._xyz); eat(m._opacity)
mask = m.dygs == False
res2 = gr.render(view, m, pipe, torch.ones(3, device="cuda"), mask=mask)
eat(res2["render"]); eat(res2["n_touched"])
print("RESULT " + json.dumps({{"binding": _C.binding(), "digest": h.hexdigest()}}))
'''


def _run(glue):
    env = dict(os.environ, GSR_GLUE=glue)
    env.pop("GSR_LIB", None)
    env.pop("GSR_EXACT_MATH", None)
    out = subprocess.run([sys.executable, "-c", SCRIPT.format(repo=REPO)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_native_and_ctypes_bindings_agree_bitwise():
    a, b = _run("native"), _run("ctypes")
    assert a["binding"] == "native" and b["binding"] == "ctypes"
    assert a["digest"] == b["digest"]


def test_cpp_autograd_node_equals_the_python_node_bitwise():
    """GaussianRasterizer's autograd node exists twice: in Python (autograd.py, the reference path) and in C++ (torch_glue.cpp RasterizeNode,
    taken when the native glue is loaded). Same calls underneath: images, radii, n_touched and every gradient must agree bit for bit --
    SH input with pose parameters of shape (3,), and precomputed colours with [1,3] pose parameters."""
    sys.path[:0] = [os.path.join(REPO, "tests"), REPO, os.path.join(REPO, "4dgs-slam_amd")]
    import numpy as np
    import torch
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    from util import make_camera, make_cotangents, make_gaussians
    if _C._glue is None or not hasattr(_C._glue, "rasterize_autograd"):
        pytest.skip("native glue not built")
    cam = make_camera(200, 152)
    g = make_gaussians(6000, cam, seed=11, sh_degree=1)
    gc, gd = make_cotangents(cam, seed=12)
    T = lambda a: torch.tensor(np.asarray(a, np.float32), device="cuda")
    rs = dgr.GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, T([0.2, 0.1, 0.3]), 1.0, T(cam.viewmatrix), T(cam.projmatrix),
                                           T(cam.projmatrix_raw), 1, T(cam.campos), False, False)

    def run(native, precomputed):
        dgr._NATIVE_NODE = native
        leaf = lambda a: T(a).requires_grad_(True)
        p = {"means3D": leaf(g["means3D"]), "opacities": leaf(g["opacities"]), "scales": leaf(g["scales"]), "rotations": leaf(g["rotations"])}
        if precomputed:
            p["colors_precomp"] = leaf(np.random.default_rng(5).uniform(0, 1, (6000, 3)))
            pose = {"theta": torch.zeros((1, 3), device="cuda", requires_grad=True), "rho": torch.zeros((1, 3), device="cuda", requires_grad=True)}
        else:
            p["shs"] = leaf(g["shs"])
            pose = {"theta": torch.zeros(3, device="cuda", requires_grad=True), "rho": torch.zeros(3, device="cuda", requires_grad=True)}
        m2d = torch.zeros((6000, 3), device="cuda", requires_grad=True)
        out = dgr.GaussianRasterizer(rs)(means2D=m2d, **p, **pose)
        torch.autograd.backward([out[0], out[2]], [T(gc), T(gd)])
        assert out[1].dtype == torch.int32 and not out[1].requires_grad and not out[4].requires_grad
        return list(out) + [m2d.grad] + [t.grad for t in p.values()] + [t.grad for t in pose.values()]

    try:
        for precomputed in (False, True):
            a, b = run(True, precomputed), run(False, precomputed)
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert x.shape == y.shape and torch.equal(x, y)
    finally:
        dgr._NATIVE_NODE = True
