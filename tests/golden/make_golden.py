#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by IMPORTING the Python pieces of the reference (run in the
authoring container only -- /root/reference does not exist on the GPU box and nothing here is read at test time
except the .npz files this script writes). The reference's CUDA extensions cannot be built or imported here, so:

  golden_ref_python.npz   outputs of importable reference helpers on seeded inputs:
                            eval_sh (gaussian_splatting/utils/sh_utils.py:57-118), build_rotation / build_scaling_rotation /
                            strip_symmetric (utils/general_utils.py:97-148), getProjectionMatrix2 / getWorld2View2
                            (utils/graphics_utils.py:33-93), SE3_exp (utils/pose_utils.py:28-77)
  golden_render_*.npz     the reference's UNMODIFIED gaussian_renderer.render()/render_flow() run end to end on CPU
                            (module stubs + a device='cuda'->'cpu' TorchFunctionMode) on top of the oracle-backed
                            stand-in for diff_gaussian_rasterization: inputs, the exact arguments that reached
                            GaussianRasterizer.forward, the output dict and the gradients of a fixed loss.
No reference source or bytecode is copied; fixtures are data only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "4dgs-slam_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        d = kwargs.get("device")
        if d is not None and "cuda" in str(d):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("tkinter", W=None)
    mod("open3d")
    mod("plyfile", PlyData=object, PlyElement=object)
    mod("simple_knn")
    mod("simple_knn._C", distCUDA2=lambda p: None)
    mod("pytorch3d")
    mod("pytorch3d.ops", knn_points=None, ball_query=None)
    mod("pytorch3d.io", load_ply=None)
    mod("pytorch3d.loss")
    mod("pytorch3d.loss.mesh_laplacian_smoothing", cot_laplacian=None)
    try:
        import cv2  # noqa: F401
    except Exception:
        mod("cv2")
    import oracle.torch_binding as ob  # the oracle-backed stand-in, under the reference's package name

    rec = {"calls": []}

    class RecordingRasterizer(ob.GaussianRasterizer):
        def forward(self, **kw):
            rec["calls"].append((self.raster_settings, {k: v for k, v in kw.items()}))
            return super().forward(**kw)

    mod("diff_gaussian_rasterization", GaussianRasterizationSettings=ob.GaussianRasterizationSettings,
        GaussianRasterizer=RecordingRasterizer)
    return rec


class Cam:
    pass


def make_cam(W, H, k=0, time=0.0):
    from synthetic_scene import make_camera, keyframe_pose
    R, t = keyframe_pose(k)
    c = make_camera(W, H, R=R, t=t)
    cam = Cam()
    T = lambda a: torch.tensor(a)
    cam.image_height, cam.image_width = H, W
    cam.FoVx, cam.FoVy = 2 * np.arctan(c.tanfovx), 2 * np.arctan(c.tanfovy)
    cam.world_view_transform, cam.full_proj_transform, cam.projection_matrix = T(c.viewmatrix), T(c.projmatrix), T(c.projmatrix_raw)
    cam.camera_center = T(c.campos)
    cam.cam_rot_delta = torch.nn.Parameter(torch.zeros(3))
    cam.cam_trans_delta = torch.nn.Parameter(torch.zeros(3))
    cam.time = time
    cam.uid = k
    return cam, c


class Model:
    """Duck-typed GaussianModel holding plain leaf tensors (activations already applied)."""

    def __init__(self, g, dygs, max_deg):
        P = g["means3D"].shape[0]
        L = lambda a: torch.tensor(a, requires_grad=True)
        self._xyz, self._features, self._opac = L(g["means3D"]), L(g["shs"]), L(g["opacities"])
        self._scal, self._rot = L(g["scales"]), L(g["rotations"])
        self.dygs = torch.tensor(dygs)
        self.active_sh_degree, self.max_sh_degree = g["sh_degree"], max_deg

    get_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: s._features)
    get_opacity = property(lambda s: s._opac)
    get_scaling = property(lambda s: s._scal)
    get_rotation = property(lambda s: s._rot)
    leaves = property(lambda s: dict(xyz=s._xyz, features=s._features, opacity=s._opac, scaling=s._scal, rotation=s._rot))


def to_np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def main():
    rec = install_stubs()
    sys.path.insert(0, REF)
    out = {}
    rng = np.random.default_rng(123)
    with CudaToCpu():
        from gaussian_splatting.utils.sh_utils import eval_sh
        from gaussian_splatting.utils.general_utils import build_rotation, build_scaling_rotation, strip_symmetric
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2, getWorld2View2
        from utils.pose_utils import SE3_exp

        # --- helper goldens ---
        N = 64
        dirs = rng.normal(size=(N, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        sh = rng.normal(size=(N, 3, 16))
        out["sh_dirs"], out["sh_coeffs"] = dirs.astype(np.float32), sh.astype(np.float32)
        for deg in range(4):
            out[f"sh_eval_deg{deg}"] = to_np(eval_sh(deg, torch.tensor(out["sh_coeffs"]), torch.tensor(out["sh_dirs"])))
        q = rng.normal(size=(N, 4)).astype(np.float32)
        s = np.exp(rng.normal(-3, 0.5, size=(N, 3))).astype(np.float32)
        out["cov_q"], out["cov_s"] = q, s
        out["cov_R"] = to_np(build_rotation(torch.tensor(q)))
        Lm = build_scaling_rotation(1.7 * torch.tensor(s), torch.tensor(q))
        out["cov_sym_mod1p7"] = to_np(strip_symmetric(Lm @ Lm.transpose(1, 2)))
        out["proj2_tum"] = to_np(getProjectionMatrix2(znear=0.01, zfar=100.0, cx=320.1, cy=247.6, fx=535.4, fy=539.2, W=640, H=480))
        from synthetic_scene import keyframe_pose
        R, t = keyframe_pose(5)
        out["w2v_R"], out["w2v_t"] = R.astype(np.float32), t.astype(np.float32)
        out["w2v"] = to_np(getWorld2View2(torch.tensor(out["w2v_R"]), torch.tensor(out["w2v_t"])))
        taus = rng.normal(0, 0.2, size=(6, 6)).astype(np.float32); taus[0] = 0; taus[1, 3:] = 1e-7
        out["se3_tau"] = taus
        out["se3_exp"] = np.stack([to_np(SE3_exp(torch.tensor(tt))) for tt in taus])
        np.savez_compressed(os.path.join(HERE, "golden_ref_python.npz"), **out)
        print("wrote golden_ref_python.npz", {k: v.shape for k, v in out.items()})

        # --- the reference wrapper end to end on the oracle stand-in ---
        import gaussian_splatting.gaussian_renderer as ref_gr
        from synthetic_scene import make_gaussians, make_cotangents

        class Pipe:
            convert_SHs_python = False
            compute_cov3D_python = False

        W, H, P = 64, 48, 300
        cam, c = make_cam(W, H, k=2, time=3.0)
        cam2, _ = make_cam(W, H, k=3, time=4.0)
        g = make_gaussians(P, c, seed=7, sh_degree=1, max_sh_degree=2, scale_mean=0.03)
        R2, t2 = keyframe_pose(2)
        g["means3D"] = ((g["means3D"].astype(np.float64) - t2) @ R2).astype(np.float32)
        dygs = rng.uniform(size=P) < 0.3
        nd = int(dygs.sum())
        gc, gd = make_cotangents(c, seed=11)
        bg = torch.tensor([1.0, 1.0, 1.0])
        d = lambda *sh_: (rng.normal(0, 0.01, size=sh_)).astype(np.float32)
        deltas = dict(dx=d(nd, 3), ds=d(nd, 3) * 0.1, dr=d(nd, 4), dx2=d(nd, 3))
        common = dict(W=W, H=H, P=P, dygs=dygs, gc=gc, gd=gd, bg=to_np(bg), **{k: v for k, v in g.items() if k != "sh_degree"},
                      sh_degree=g["sh_degree"], max_sh_degree=2, cam_k=2, cam2_k=3, **deltas)

        def run(tag, fn):
            rec["calls"].clear()
            m = Model(g, dygs, 2)
            for cm in (cam, cam2):
                cm.cam_rot_delta.grad = None; cm.cam_trans_delta.grad = None
            extra = {}
            res = fn(m, extra)
            loss = (res["render"] * torch.tensor(gc)).sum() + (res["depth"] * torch.tensor(gd)).sum()
            loss.backward()
            (rs, kw), = rec["calls"]
            fx = dict(common)
            for k, v in kw.items():
                fx["arg_" + k] = to_np(v) if v is not None else np.zeros(0, np.float32)
                fx["argrg_" + k] = np.array(bool(getattr(v, "requires_grad", False)))
                fx["argnone_" + k] = np.array(v is None)
            for k in rs._fields:
                fx["rs_" + k] = to_np(getattr(rs, k))
            for k, v in res.items():
                fx["out_" + k] = to_np(v)
            for k, v in m.leaves.items():
                fx["grad_" + k] = to_np(v.grad) if v.grad is not None else np.zeros(0, np.float32)
            fx["grad_viewspace"] = to_np(res["viewspace_points"].grad)
            fx["grad_theta"] = to_np(cam.cam_rot_delta.grad) if cam.cam_rot_delta.grad is not None else np.zeros(0, np.float32)
            fx["grad_rho"] = to_np(cam.cam_trans_delta.grad) if cam.cam_trans_delta.grad is not None else np.zeros(0, np.float32)
            for k, v in extra.items():
                fx["extra_" + k] = to_np(v.grad) if isinstance(v, torch.Tensor) and v.grad is not None else to_np(v)
            np.savez_compressed(os.path.join(HERE, f"golden_render_{tag}.npz"), **fx)
            print("wrote", tag, "R visible", int((res["radii"] > 0).sum()))

        pipe = Pipe()
        # the call shapes used by the reference callers (SURVEY.md 8b last row)
        run("static", lambda m, ex: ref_gr.render(cam, m, pipe, bg, dx=0, ds=0, dr=None))                       # slam_backend.py:1038-1044
        run("tracking", lambda m, ex: ref_gr.render(cam, m, pipe, bg, dx=None, ds=None, dr=None, mask=(m.dygs == False)))  # slam_frontend.py:412-414
        def dyn(m, ex):
            ex["dx"] = torch.tensor(deltas["dx"], requires_grad=True); ex["ds"] = torch.tensor(deltas["ds"], requires_grad=True)
            ex["dr"] = torch.tensor(deltas["dr"], requires_grad=True)
            return ref_gr.render(cam, m, pipe, bg, dx=ex["dx"], ds=ex["ds"], dr=ex["dr"])                          # slam_backend.py:397-408
        run("dynamic", dyn)
        run("eval", lambda m, ex: ref_gr.render(cam, m, pipe, bg, dx=0, dr=0, ds=0))                              # eval_utils.py:339-344
        def flow(m, ex):
            ex["dx"] = torch.tensor(deltas["dx"], requires_grad=True); ex["dx2"] = torch.tensor(deltas["dx2"], requires_grad=True)
            ex["ds"] = torch.tensor(deltas["ds"], requires_grad=True); ex["dr"] = torch.tensor(deltas["dr"], requires_grad=True)
            r = ref_gr.render_flow(m, cam, cam2, ex["dx"], ex["dx2"], ex["dr"], ex["ds"])
            return r
        run("flow", flow)


if __name__ == "__main__":
    main()
