"""Synthetic scene + camera generator shared by bench.py and the tests.

Numpy only. Matrix conventions follow the reference callers:
  viewmatrix     = W2C.T                      (utils/camera_utils.py:124-126)
  projmatrix_raw = getProjectionMatrix2(...).T (utils/slam_frontend.py:615-624,
                                               gaussian_splatting/utils/graphics_utils.py:72-93)
  projmatrix     = W2C.T @ P.T                (utils/camera_utils.py:128-134)
  campos         = inverse(W2C.T)[3,:3]       (utils/camera_utils.py:146-148)
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import types

import numpy as np

# TUM fr3 intrinsics, configs/rgbd/tum/fr3_sitting_static.yaml:7-18
TUM_FR3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, W=640, H=480)
C0 = 0.28209479177387814  # gaussian_splatting/utils/sh_utils.py RGB2SH


def projection_matrix2(znear, zfar, cx, cy, fx, fy, W, H) -> np.ndarray:
    """4x4 OpenCV-style projection used by the reference (graphics_utils.py:72-93), un-transposed."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left, right = znear / fx * left, znear / fx * right
    top, bottom = znear / fy * top, znear / fy * bottom
    P = np.zeros((4, 4), np.float64)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class CameraArrays:
    W: int
    H: int
    fx: float
    fy: float
    cx: float
    cy: float
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray      # [4,4] f32, = W2C.T
    projmatrix: np.ndarray      # [4,4] f32, = W2C.T @ P.T
    projmatrix_raw: np.ndarray  # [4,4] f32, = P.T
    campos: np.ndarray          # [3] f32


def make_camera(W=640, H=480, fx=None, fy=None, cx=None, cy=None, R=None, t=None, znear=0.01, zfar=100.0) -> CameraArrays:
    s = W / 640.0
    fx = TUM_FR3["fx"] * s if fx is None else fx
    fy = TUM_FR3["fy"] * (H / 480.0) if fy is None else fy
    cx = TUM_FR3["cx"] * s if cx is None else cx
    cy = TUM_FR3["cy"] * (H / 480.0) if cy is None else cy
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    t = np.zeros(3) if t is None else np.asarray(t, np.float64)
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = t
    P = projection_matrix2(znear, zfar, cx, cy, fx, fy, W, H)
    view = w2c.T
    proj_raw = P.T
    full = view @ proj_raw
    campos = np.linalg.inv(view)[3, :3]
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return CameraArrays(W, H, fx, fy, cx, cy, W / (2 * fx), H / (2 * fy), f32(view), f32(full), f32(proj_raw), f32(campos))


def keyframe_pose(k: int):
    """Keyframe k of SURVEY 8d: rotation about y by 0.02*k rad, translation (0.03*k, 0, 0)."""
    a = 0.02 * k
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    return R, np.array([0.03 * k, 0.0, 0.0])


def make_gaussians(P: int, cam: CameraArrays, seed: int = 0, sh_degree: int = 0, max_sh_degree: int | None = None,
                   scale_mean: float = 0.005) -> dict:
    """Frustum-filling random Gaussians. Returns float32 numpy arrays in the rasterizer's input layout."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.5, 6.0, P)
    x = z * cam.tanfovx * rng.uniform(-1.05, 1.05, P)
    y = z * cam.tanfovy * rng.uniform(-1.05, 1.05, P)
    means = np.stack([x, y, z], 1)
    scales = np.exp(rng.normal(math.log(scale_mean), 0.5, (P, 3)))
    q = rng.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, (P, 1))))
    max_deg = sh_degree if max_sh_degree is None else max_sh_degree
    M = (max_deg + 1) ** 2
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = (rng.uniform(0, 1, (P, 3)) - 0.5) / C0
    if M > 1:
        shs[:, 1:, :] = rng.normal(0, 0.1, (P, M - 1, 3))
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return dict(means3D=f32(means), scales=f32(scales), rotations=f32(q), opacities=f32(opac), shs=f32(shs),
                sh_degree=sh_degree)


def make_cotangents(cam: CameraArrays, seed: int = 1):
    """dL/dcolor = N(0,1)/(3N), dL/ddepth = N(0,1)/N."""
    rng = np.random.default_rng(seed)
    N = cam.W * cam.H
    gc = rng.normal(0, 1, (3, cam.H, cam.W)) / (3 * N)
    gd = rng.normal(0, 1, (1, cam.H, cam.W)) / N
    return np.ascontiguousarray(gc, np.float32), np.ascontiguousarray(gd, np.float32)


class GaussianModelStub:
    """The attribute surface of scene/gaussian_model.py that render() touches (raw leaves + activations, :60-68,100-128), filled from a
    make_gaussians() scene: what the benches and tests hand to gaussian_renderer.render() in place of a GaussianModel."""

    def __init__(self, g, isotropic, dyn_frac, seed):
        import torch
        dev = "cuda"
        L = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=True)
        rng = np.random.default_rng(seed)
        P = g["means3D"].shape[0]
        self._xyz = L(g["means3D"])
        sc = np.log(g["scales"])
        self._scaling = L(sc[:, :1] if isotropic else sc)
        self._rotation = L(g["rotations"] * rng.uniform(0.5, 2.0, size=(P, 1)))          # not unit length: normalize matters
        op = np.clip(g["opacities"].reshape(P, 1), 1e-4, 1 - 1e-4)
        self._opacity = L(np.log(op / (1 - op)))
        self._features_dc = L(g["shs"][:, :1])
        self._features_rest = L(g["shs"][:, 1:])
        self.active_sh_degree = g.get("sh_degree", 0)
        self.max_sh_degree = 3
        self.dygs = torch.tensor(rng.uniform(size=P) < dyn_frac, device=dev)
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    get_features = property(lambda s: __import__("torch").cat((s._features_dc, s._features_rest), dim=1))
    leaves = property(lambda s: dict(xyz=s._xyz, scaling=s._scaling, rotation=s._rotation, opacity=s._opacity,
                                     f_dc=s._features_dc, f_rest=s._features_rest))


def camera_namespace(cam):
    import torch
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
    return types.SimpleNamespace(
        FoVx=2 * np.arctan(cam.tanfovx), FoVy=2 * np.arctan(cam.tanfovy), image_height=cam.H, image_width=cam.W,
        world_view_transform=T(cam.viewmatrix), full_proj_transform=T(cam.projmatrix), projection_matrix=T(cam.projmatrix_raw),
        camera_center=T(cam.campos), cam_rot_delta=T(np.zeros(3), True), cam_trans_delta=T(np.zeros(3), True), time=0.0)
