"""Camera / keyframe object of the SLAM loop -- the counterpart of the reference's ``utils/camera_utils.py`` Camera (:21-148, 205-233,
438-451) without the RAFT / YOLO members (generate_flow, render_mask: external networks, out of scope) and of
``utils/pose_utils.py`` (update_pose, :80-97).

Difference by design: the pose [R | T], the camera deltas and the three matrices the rasterizer reads (``world_view_transform``,
``full_proj_transform``, ``camera_center``) live in PERSISTENT device buffers that ``gsr_camera_step_launch`` (include/slam_map.h)
updates in place with one launch and no host synchronisation. The reference recomputes the matrices with a handful of torch
kernels (and a 4x4 inverse) in every property access, i.e. several times per rendered view; stable addresses are also what lets
a tracking iteration be captured in a hipGraph (slam/tracking_graph.py)."""
import ctypes as C
import math

import numpy as np
import torch
from torch import nn

from . import _lib


def image_gradient(image):
    """Scharr gradients (utils/slam_utils.py:5-22). The reference runs two grouped conv2d; the 3x3 stencils are written out on shifted
    views here (MIOpen spends ~17 ms per call choosing a convolution for this one-channel 3x3 problem -- 84 ms per frame)."""
    p = torch.nn.functional.pad(image, (1, 1, 1, 1), mode="reflect")
    t, m, b = p[:, :-2], p[:, 1:-1], p[:, 2:]                 # rows y-1, y, y+1
    l, c, r = slice(0, -2), slice(1, -1), slice(2, None)      # columns x-1, x, x+1
    normalizer = 1.0 / 16.0
    img_grad_v = normalizer * ((3 * t[..., l] + 10 * t[..., c] + 3 * t[..., r]) - (3 * b[..., l] + 10 * b[..., c] + 3 * b[..., r]))
    img_grad_h = normalizer * ((3 * t[..., l] + 10 * m[..., l] + 3 * b[..., l]) - (3 * t[..., r] + 10 * m[..., r] + 3 * b[..., r]))
    return img_grad_v, img_grad_h


def image_gradient_mask(image, eps=0.01):
    """utils/slam_utils.py:25-39: true where all nine pixels of the 3x3 neighbourhood exceed eps."""
    p = torch.abs(torch.nn.functional.pad(image, (1, 1, 1, 1), mode="reflect")) > eps
    rows = p[:, :-2] & p[:, 1:-1] & p[:, 2:]
    full = rows[..., :-2] & rows[..., 1:-1] & rows[..., 2:]
    return full, full


def compute_grad_mask_hip(original_image, edge_threshold):
    """The non-replica branch of compute_grad_mask as three HIP launches (gsr_edge_mask, include/slam_map.h) instead of ~25 torch
    launches (11 ms of host time per frame at 640x480). Returns the bool mask [1,H,W]."""
    img = original_image if (original_image.dtype == torch.float32 and original_image.is_contiguous()) else original_image.float().contiguous()
    _, H, W = img.shape
    scratch = torch.empty((H * W + 1,), dtype=torch.float32, device=img.device)
    mask = torch.empty((1, H, W), dtype=torch.uint8, device=img.device)
    L = _lib.lib()
    with torch.cuda.device(img.device):
        rc = L.gsr_edge_mask(img.data_ptr(), H, W, float(edge_threshold), 0.01, scratch.data_ptr(), scratch[H * W:].data_ptr(), mask.data_ptr(),
                             _lib.stream(img.device))
    _lib.check(rc, "gsr_edge_mask")
    return mask.view(torch.bool)


def compute_grad_mask(original_image, config):
    """utils/camera_utils.py:205-233: pixels whose Scharr gradient magnitude exceeds edge_threshold x the image median (the non-replica
    branch is the shipped TUM / Bonn configuration; replica's per-block medians are kept as in the reference). Device images take the
    fused kernels; this tensor program is what they are tested against (and what runs on CPU tensors)."""
    edge_threshold = config["Training"]["edge_threshold"]
    if original_image.is_cuda and config["Dataset"]["type"] != "replica" and original_image.shape[0] == 3:
        return compute_grad_mask_hip(original_image, edge_threshold)
    gray_img = original_image.mean(dim=0, keepdim=True)
    gray_grad_v, gray_grad_h = image_gradient(gray_img)
    mask_v, mask_h = image_gradient_mask(gray_img)
    gray_grad_v, gray_grad_h = gray_grad_v * mask_v, gray_grad_h * mask_h
    img_grad_intensity = torch.sqrt(gray_grad_v ** 2 + gray_grad_h ** 2)
    if config["Dataset"]["type"] == "replica":
        row, col = 32, 32
        _, h, w = original_image.shape
        for r in range(row):
            for c in range(col):
                block = img_grad_intensity[:, r * int(h / row):(r + 1) * int(h / row), c * int(w / col):(c + 1) * int(w / col)]
                th_median = block.median()
                block[block > (th_median * edge_threshold)] = 1
                block[block <= (th_median * edge_threshold)] = 0
        return img_grad_intensity
    flat = img_grad_intensity.reshape(-1)                 # the median through a sort: see frontend.lower_median
    return img_grad_intensity > torch.sort(flat)[0][(flat.numel() - 1) // 2] * edge_threshold


def getProjectionMatrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """gaussian_splatting/utils/graphics_utils.py:72-93 (un-transposed; callers transpose, utils/slam_frontend.py:615-624)."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left, right = znear / fx * left, znear / fx * right
    top, bottom = znear / fy * top, znear / fy * bottom
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera(nn.Module):
    def __init__(self, uid, color, depth, gt_T, projection_matrix, fx, fy, cx, cy, fovx, fovy, image_height, image_width, time,
                 motion_mask=None, device="cuda:0"):
        super().__init__()
        self.uid = uid
        self.device = torch.device(device)
        dev = self.device
        # persistent pose + matrices (see module docstring)
        self._R = torch.eye(3, device=dev)
        self._T = torch.zeros(3, device=dev)
        self._view = torch.zeros(4, 4, device=dev)
        self._full = torch.zeros(4, 4, device=dev)
        self._campos = torch.zeros(3, device=dev)
        self._converged = torch.zeros(1, dtype=torch.int32, device=dev)
        self._adam = torch.zeros(17, device=dev)          # exp_avg[8] | exp_avg_sq[8] | step
        gt_T = torch.as_tensor(gt_T, dtype=torch.float32, device=dev)
        self.R_gt = gt_T[:3, :3].contiguous()
        self.T_gt = gt_T[:3, 3].contiguous()

        self.original_image = color
        self.depth = depth                 # numpy [H,W] like the reference's dataset output (utils/dataset.py); None for monocular
        self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        self.time = time
        self.fid = torch.tensor([time], dtype=torch.float32, device=dev)
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=dev))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=dev))
        self.exposure_a = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=dev))
        self.exposure_b = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=dev))
        self.projection_matrix = projection_matrix.to(device=dev, dtype=torch.float32).contiguous()
        self.motion_mask = motion_mask
        self._depth_dev = None
        self.refresh_matrices()

    # ---- construction -------------------------------------------------------------------------------------------
    @staticmethod
    def init_from_dataset(dataset, idx, projection_matrix):
        """utils/camera_utils.py:93-117."""
        gt_color, gt_depth, gt_pose, motion_mask = dataset[idx]
        time = idx / max(dataset.num_imgs - 1, 1)
        return Camera(idx, gt_color, gt_depth, gt_pose, projection_matrix, dataset.fx, dataset.fy, dataset.cx, dataset.cy, dataset.fovx,
                      dataset.fovy, dataset.height, dataset.width, time, motion_mask, device=dataset.device)

    # ---- pose ---------------------------------------------------------------------------------------------------
    @property
    def R(self):
        return self._R

    @property
    def T(self):
        return self._T

    def update_RT(self, R, t):
        """utils/camera_utils.py:149-151, in place on the persistent buffers, then the matrices are refreshed (one launch)."""
        self._R.copy_(torch.as_tensor(R, dtype=torch.float32).to(self.device))
        self._T.copy_(torch.as_tensor(t, dtype=torch.float32).to(self.device))
        self.refresh_matrices()

    def _step_desc(self, grads, lrs, do_pose, threshold, latch=False):
        d = _lib.CameraStep()
        ptr = lambda t: None if t is None else t.data_ptr()
        d.rot_delta, d.trans_delta = ptr(self.cam_rot_delta), ptr(self.cam_trans_delta)
        d.exposure_a, d.exposure_b = ptr(self.exposure_a), ptr(self.exposure_b)
        g = grads or {}
        d.g_rot_delta, d.g_trans_delta = ptr(g.get("rot")), ptr(g.get("trans"))
        d.g_exposure_a, d.g_exposure_b = ptr(g.get("a")), ptr(g.get("b"))
        d.exp_avg, d.exp_avg_sq, d.step = self._adam.data_ptr(), self._adam[8:].data_ptr(), self._adam[16:].data_ptr()
        d.lr_rot, d.lr_trans, d.lr_exposure = lrs
        d.beta1, d.beta2, d.eps = 0.9, 0.999, 1e-8          # torch.optim.Adam defaults, as utils/slam_frontend.py:376 uses them
        d.R, d.T, d.projmatrix = self._R.data_ptr(), self._T.data_ptr(), self.projection_matrix.data_ptr()
        d.viewmatrix, d.full_proj, d.campos = self._view.data_ptr(), self._full.data_ptr(), self._campos.data_ptr()
        d.converged, d.converged_threshold, d.do_pose = self._converged.data_ptr(), float(threshold), int(do_pose)
        d.latch = int(bool(latch))
        return d

    def refresh_matrices(self):
        d = self._step_desc(None, (0.0, 0.0, 0.0), False, 0.0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gsr_camera_step_launch(C.byref(d), _lib.stream(self.device)), "gsr_camera_step_launch")

    def reset_pose_optimizer(self):
        """A fresh torch.optim.Adam per tracked frame / per keyframe window (utils/slam_frontend.py:376, slam_backend.py:992)."""
        self._adam.zero_()
        self._converged.zero_()

    @torch.no_grad()
    def pose_step(self, lr_rot, lr_trans, lr_exposure=0.01, optimize_pose=True, optimize_exposure=True, converged_threshold=1e-4, latch=False):
        """pose_optimizer.step() + zero_grad() + update_pose(viewpoint) of the tracking / mapping loops (utils/slam_frontend.py:434-440,
        utils/slam_backend.py:748-755) in ONE launch; the gradients are read from .grad and cleared. Returns nothing: read
        ``converged()`` when the host needs the flag (that is the only synchronisation). latch=True (tracking): once the flag is set, further
        steps of the same frame do nothing, so polling it every few iterations ends at the pose the reference's immediate break leaves."""
        g = {}
        if optimize_pose and self.cam_rot_delta.grad is not None and self.cam_trans_delta.grad is not None:
            g["rot"], g["trans"] = self.cam_rot_delta.grad, self.cam_trans_delta.grad
        if optimize_exposure and self.exposure_a.grad is not None and self.exposure_b.grad is not None:
            g["a"], g["b"] = self.exposure_a.grad, self.exposure_b.grad
        d = self._step_desc(g, (float(lr_rot), float(lr_trans), float(lr_exposure)), optimize_pose, converged_threshold, latch)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gsr_camera_step_launch(C.byref(d), _lib.stream(self.device)), "gsr_camera_step_launch")
        for p in (self.cam_rot_delta, self.cam_trans_delta, self.exposure_a, self.exposure_b):
            p.grad = None

    def converged(self) -> bool:
        return bool(self._converged.item())

    @staticmethod
    @torch.no_grad()
    def pose_steps(requests):
        """pose_step() of several cameras in ONE launch (gsr_camera_steps_launch): requests = [(camera, lr_rot, lr_trans, lr_exposure,
        optimize_pose, optimize_exposure)]. The window keyframes of a mapping iteration are stepped together (utils/slam_backend.py:748-755,
        :1213-1222); per camera the arithmetic is pose_step's."""
        requests = list(requests)
        for lo in range(0, len(requests), _lib.CAMERA_STEPS_MAX):
            part = requests[lo:lo + _lib.CAMERA_STEPS_MAX]
            arr = (_lib.CameraStep * len(part))()
            for k, (cam, lr_rot, lr_trans, lr_exposure, optimize_pose, optimize_exposure) in enumerate(part):
                g = {}
                if optimize_pose and cam.cam_rot_delta.grad is not None and cam.cam_trans_delta.grad is not None:
                    g["rot"], g["trans"] = cam.cam_rot_delta.grad, cam.cam_trans_delta.grad
                if optimize_exposure and cam.exposure_a.grad is not None and cam.exposure_b.grad is not None:
                    g["a"], g["b"] = cam.exposure_a.grad, cam.exposure_b.grad
                arr[k] = cam._step_desc(g, (float(lr_rot), float(lr_trans), float(lr_exposure)), optimize_pose, 1e-4, False)
            dev = part[0][0].device
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().gsr_camera_steps_launch(len(part), arr, _lib.stream(dev)), "gsr_camera_steps_launch")
            for cam, *_ in part:
                for p in (cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b):
                    p.grad = None

    # ---- matrices the renderer reads (utils/camera_utils.py:124-148) -----------------------------------------------------
    @property
    def world_view_transform(self):
        return self._view

    @property
    def full_proj_transform(self):
        return self._full

    @property
    def camera_center(self):
        return self._campos

    # ---- keyframe data ----------------------------------------------------------------------------------------------
    def depth_device(self):
        if self._depth_dev is None and self.depth is not None:
            self._depth_dev = torch.as_tensor(self.depth, dtype=torch.float32, device=self.device).contiguous()
        return self._depth_dev

    def compute_grad_mask(self, config):
        self.grad_mask = compute_grad_mask(self.original_image, config)

    def clean(self):
        """utils/camera_utils.py:438-448; also drops the loss cache of slam_losses (it kept ~10 MB per tracked frame alive)."""
        self.original_image = None
        self.depth = None
        self._depth_dev = None
        self.grad_mask = None
        self.motion_mask = None
        self.cam_rot_delta = None
        self.cam_trans_delta = None
        self.exposure_a = None
        self.exposure_b = None
        import slam_losses
        slam_losses.drop_keyframe_constants(self)

    def clean_key(self):
        self.grad_mask = None


# ---- utils/pose_utils.py, tensor form (kept for callers that hold plain tensors; the loops use Camera.pose_step) ----------------
def skew_sym_mat(x):
    ssm = torch.zeros(3, 3, device=x.device, dtype=x.dtype)
    ssm[0, 1], ssm[0, 2], ssm[1, 0], ssm[1, 2], ssm[2, 0], ssm[2, 1] = -x[2], x[1], x[2], -x[0], -x[1], x[0]
    return ssm


def SO3_exp(theta):
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + W + 0.5 * W2
    return I + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / (angle ** 2)) * W2


def V(theta):
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    if angle < 1e-5:
        return I + 0.5 * W + (1.0 / 6.0) * W2
    return I + W * ((1.0 - torch.cos(angle)) / (angle ** 2)) + W2 * ((angle - torch.sin(angle)) / (angle ** 3))


def SE3_exp(tau):
    rho, theta = tau[:3], tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V(theta) @ rho
    return T


def update_pose(camera, converged_threshold=1e-4):
    """utils/pose_utils.py:80-97 with the reference's return value (host bool: synchronises). The SLAM loops call
    ``camera.pose_step`` instead, which fuses the optimizer step and leaves the flag on the device."""
    camera.pose_step(0.0, 0.0, optimize_pose=True, optimize_exposure=False, converged_threshold=converged_threshold)
    return camera.converged()


def fov_from_focal(f, size):
    return 2 * math.atan(size / (2 * f))
