"""Synthetic RGB-D sequence for the SLAM loop (BASELINE config #4 without external assets): a ground-truth Gaussian scene (a room:
back wall, floor, side walls, a few boxes; optionally one moving object) rendered along a smooth camera trajectory by THIS
repository's rasterizer. It has the interface the reference's loops expect from ``utils/dataset.py`` datasets:

    dataset[idx] -> (gt_color [3,H,W] float cuda, gt_depth [H,W] float32 numpy, gt_pose [4,4] W2C tensor, motion_mask [H,W] bool cuda)
    dataset.fx / fy / cx / cy / fovx / fovy / width / height / num_imgs / device / dynamic_objects / dystart

``motion_mask`` follows the reference's convention: True = static pixel (``~motion_mask`` selects the moving object,
utils/slam_backend.py:486-488). The TUM / Bonn loaders, YOLO masks and RAFT flow of the reference are out of scope (external data and
weights); the ground-truth flow this generator can produce stands in for RAFT where the dynamic branch wants one."""
import math

import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

from .camera import fov_from_focal, getProjectionMatrix2

C0 = 0.28209479177387814


def _texture(p, rng_phase):
    """Smooth + mid-frequency procedural colour in [0.05, 0.95] from 3-D position (tracking needs gradients at several scales)."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    a, b, c = rng_phase
    r = 0.5 + 0.25 * np.sin(2.1 * x + a) * np.cos(1.7 * y + b) + 0.2 * np.sin(9.0 * x + 7.0 * y + 5.0 * z + c)
    g = 0.5 + 0.25 * np.sin(1.3 * y + 2.0 * z + b) + 0.2 * np.cos(8.0 * x - 6.0 * y + 4.0 * z + a)
    bch = 0.5 + 0.25 * np.cos(1.9 * z + 1.1 * x + c) + 0.2 * np.sin(7.0 * x + 9.0 * z - 5.0 * y + b)
    checker = ((np.floor(x * 3.0) + np.floor(y * 3.0) + np.floor(z * 3.0)) % 2) * 0.12 - 0.06
    col = np.stack([r + checker, g - checker, bch + checker], 1)
    return np.clip(col, 0.05, 0.95)


def _plane(origin, u, v, nu, nv, spacing, rng):
    """Jittered grid of points on the parallelogram origin + s u + t v."""
    s, t = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    s = (s.reshape(-1) + rng.uniform(-0.3, 0.3, nu * nv)) * spacing
    t = (t.reshape(-1) + rng.uniform(-0.3, 0.3, nu * nv)) * spacing
    return origin[None] + s[:, None] * u[None] + t[:, None] * v[None]


def build_room(spacing=0.03, seed=0):
    """Static scene points [P,3] (camera looks down +z, y down like the TUM convention) and colours [P,3]."""
    rng = np.random.default_rng(seed)
    ex, ey, ez = np.eye(3)
    parts = []
    n = lambda length: int(length / spacing) + 1
    parts.append(_plane(np.array([-2.4, -1.7, 3.2]), ex, ey, n(4.8), n(3.0), spacing, rng))          # back wall z = 3.2
    parts.append(_plane(np.array([-2.4, 1.3, 0.6]), ex, ez, n(4.8), n(2.6), spacing, rng))           # floor y = 1.3
    parts.append(_plane(np.array([-2.4, -1.7, 0.6]), ez, ey, n(2.6), n(3.0), spacing, rng))          # left wall x = -2.4
    parts.append(_plane(np.array([2.4, -1.7, 0.6]), ez, ey, n(2.6), n(3.0), spacing, rng))           # right wall x = 2.4
    for (cx, cy, cz, sx, sy, sz) in ((-0.9, 0.8, 2.3, 0.7, 0.5, 0.5), (0.9, 0.6, 2.6, 0.6, 0.7, 0.4), (0.1, 0.95, 1.9, 0.5, 0.35, 0.5)):
        o = np.array([cx - sx / 2, cy - sy / 2, cz - sz / 2])
        parts.append(_plane(o, ex, ey, n(sx), n(sy), spacing, rng))                                  # front face
        parts.append(_plane(o, ex, ez, n(sx), n(sz), spacing, rng))                                  # top face
        parts.append(_plane(o, ez, ey, n(sz), n(sy), spacing, rng))                                  # left face
        parts.append(_plane(o + np.array([sx, 0, 0]), ez, ey, n(sz), n(sy), spacing, rng))           # right face
    pts = np.concatenate(parts, 0)
    return pts.astype(np.float32), _texture(pts, rng.uniform(0, 6.28, 3)).astype(np.float32)


def build_ball(center, radius=0.22, spacing=0.03, seed=1):
    rng = np.random.default_rng(seed)
    n = int(4 * math.pi * radius * radius / (spacing * spacing))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.asarray(center)[None] + radius * d
    col = np.clip(np.stack([0.85 + 0.1 * np.sin(20 * d[:, 0]), 0.25 + 0.15 * np.sin(17 * d[:, 1]), 0.2 + 0.1 * d[:, 2]], 1), 0.05, 0.95)
    return pts.astype(np.float32), col.astype(np.float32)


def trajectory(num_frames, step=0.006, rot_step=0.0035):
    """World-to-camera poses [num_frames,4,4]: a gentle arc (translation ~ step per frame, yaw/pitch ~ rot_step rad per frame)."""
    poses = []
    for k in range(num_frames):
        yaw, pitch = rot_step * k * math.cos(0.03 * k), 0.4 * rot_step * k * math.sin(0.05 * k)
        Ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
        R = Rx @ Ry
        c = np.array([step * k * math.cos(0.04 * k), 0.35 * step * k * math.sin(0.07 * k), 0.5 * step * k])     # camera centre (world)
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ c
        poses.append(T)
    return np.stack(poses).astype(np.float32)


class SyntheticRGBDDataset:
    def __init__(self, num_frames=40, width=320, height=240, seed=0, dynamic=False, dystart=None, spacing=0.03, device="cuda:0",
                 step=0.006, rot_step=0.0035, depth_noise=0.0):
        self.device = torch.device(device)
        self.num_imgs, self.width, self.height = num_frames, width, height
        s = width / 640.0
        self.fx, self.fy, self.cx, self.cy = 535.4 * s, 539.2 * (height / 480.0), 320.1 * s, 247.6 * (height / 480.0)     # TUM fr3
        self.fovx, self.fovy = fov_from_focal(self.fx, width), fov_from_focal(self.fy, height)
        self.dynamic = bool(dynamic)
        self.dystart = (0 if dystart is None else int(dystart)) if dynamic else num_frames + 1     # first frame with the moving object
        self.dynamic_objects = 0
        self.poses = torch.tensor(trajectory(num_frames, step, rot_step), device=self.device)
        pts, col = build_room(spacing, seed)
        self.static_xyz, self.static_rgb = torch.tensor(pts, device=self.device), torch.tensor(col, device=self.device)
        self.ball_xyz0, self.ball_rgb = (torch.tensor(a, device=self.device) for a in build_ball((-0.5, 0.2, 1.9), spacing=spacing, seed=seed + 1))
        self.spacing = spacing
        self.depth_noise = depth_noise
        self._rng = torch.Generator(device="cpu").manual_seed(seed + 7)
        self.projection_matrix = getProjectionMatrix2(0.01, 100.0, self.cx, self.cy, self.fx, self.fy, width, height).transpose(0, 1).to(self.device)
        self._cache = {}
        self._flow_cache = {}

    def __len__(self):
        return self.num_imgs

    # ---- ground truth ---------------------------------------------------------------------------------------------
    def ball_offset(self, idx):
        """World-space displacement of the moving object at frame idx (zero before dystart)."""
        k = max(0, idx - self.dystart)
        return torch.tensor([0.012 * k, -0.004 * k, 0.003 * k], device=self.device)

    def gt_gaussians(self, idx, with_ball=None):
        with_ball = (self.dynamic and idx >= self.dystart) if with_ball is None else with_ball
        xyz, rgb = self.static_xyz, self.static_rgb
        if with_ball:
            xyz = torch.cat([xyz, self.ball_xyz0 + self.ball_offset(idx)], 0)
            rgb = torch.cat([rgb, self.ball_rgb], 0)
        return xyz, rgb

    def _raster(self, xyz, rgb, pose, bg):
        P, dev = xyz.shape[0], self.device
        view = pose.transpose(0, 1).contiguous()
        full = (view @ self.projection_matrix).contiguous()
        campos = torch.linalg.inv(view)[3, :3].contiguous()
        rs = GaussianRasterizationSettings(image_height=self.height, image_width=self.width, tanfovx=math.tan(self.fovx * 0.5),
                                           tanfovy=math.tan(self.fovy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=view, projmatrix=full,
                                           projmatrix_raw=self.projection_matrix, sh_degree=0, campos=campos, prefiltered=False, debug=False)
        scales = torch.full((P, 3), 0.62 * self.spacing, device=dev)
        rots = torch.zeros((P, 4), device=dev)
        rots[:, 0] = 1
        opac = torch.full((P, 1), 0.97, device=dev)
        shs = ((rgb - 0.5) / C0)[:, None, :].contiguous()
        with torch.no_grad():
            color, _, depth, opacity, _ = GaussianRasterizer(rs)(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=opac, shs=shs,
                                                                 scales=scales, rotations=rots)
        return color, depth, opacity

    def __getitem__(self, idx):
        if idx in self._cache:
            return self._cache[idx]
        pose = self.poses[idx]
        bg = torch.zeros(3, device=self.device)
        xyz, rgb = self.gt_gaussians(idx)
        color, depth, opacity = self._raster(xyz, rgb, pose, bg)
        d = torch.where(opacity > 0.98, depth / opacity.clamp_min(1e-6), torch.zeros_like(depth))[0]      # expected depth where the surface is opaque
        if self.depth_noise > 0:
            d = d * (1.0 + self.depth_noise * torch.randn(d.shape, generator=self._rng).to(d.device)) * (d > 0)
        motion = torch.ones((self.height, self.width), dtype=torch.bool, device=self.device)
        if self.dynamic and idx >= self.dystart:
            _, _, a = self._raster(self.ball_xyz0 + self.ball_offset(idx), self.ball_rgb, pose, bg)
            # the ball may be hidden behind static geometry: it is "moving pixels" only where it is the visible surface
            _, ds, os_ = self._raster(self.static_xyz, self.static_rgb, pose, bg)
            ball_front = (a[0] > 0.3)
            motion = ~ball_front
            self.dynamic_objects = 1
        item = (color.clamp(0, 1).contiguous(), d.cpu().numpy().astype(np.float32), pose.clone(), motion)
        self._cache[idx] = item
        return item

    def gt_flow(self, idx_from, idx_to):
        """NDC flow [H,W,2] of frame idx_from's surface points into frame idx_to (what the reference asks RAFT for, scaled as
        utils/camera_utils.py:412-413 does: pixels / (W, H) * 2), plus a validity mask. Cached per pair, like the reference keeps RAFT's
        output on the keyframe."""
        hit = self._flow_cache.get((idx_from, idx_to))
        if hit is not None:
            return hit
        color, depth, pose, motion = self[idx_from]
        H, W, dev = self.height, self.width, self.device
        d = torch.as_tensor(depth, device=dev)
        v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        pc = torch.stack([(u - self.cx) / self.fx * d, (v - self.cy) / self.fy * d, d, torch.ones_like(d)], -1)          # camera frame
        pw = pc @ torch.linalg.inv(pose).transpose(0, 1)
        if self.dynamic:
            delta = self.ball_offset(idx_to) - self.ball_offset(idx_from)
            pw = pw + torch.cat([delta, delta.new_zeros(1)])[None, None] * (~motion)[..., None]
        p2 = pw @ self.poses[idx_to].transpose(0, 1)
        u2, v2 = p2[..., 0] / p2[..., 2] * self.fx + self.cx, p2[..., 1] / p2[..., 2] * self.fy + self.cy
        flow = torch.stack([(u2 - u) / W * 2, (v2 - v) / H * 2], -1)
        self._flow_cache[(idx_from, idx_to)] = (flow, d > 0)
        return flow, d > 0
