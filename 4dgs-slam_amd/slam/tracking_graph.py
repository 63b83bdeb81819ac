"""A tracking iteration as ONE hipGraph launch (utils/slam_frontend.py:405-448 is the loop being replaced):

    render(static Gaussians) -> fused tracking loss -> backward -> camera step (Adam on pose + exposure, update_pose, matrices)

Eager, that is ~12 kernel launches plus PyTorch's autograd bookkeeping: ~0.2 ms of host time per iteration, which is what bounds
scenes below ~50k Gaussians (100 iterations per frame). Captured, the host does one ``graph.replay()`` per iteration and reads one
flag every few iterations. Three things make the iteration capturable:
  * the rasterizer's lazy mode (gsr_set_option("lazy", 1)): the forward pass does not wait for num_rendered;
  * the camera lives in persistent device buffers and its optimizer step / pose update / matrix refresh is one kernel whose Adam
    step counter is in device memory (Camera.pose_step, include/slam_map.h);
  * the frame's data (image, depth, loss weights, pose) is COPIED into a static slot per frame, so one captured graph serves every
    frame tracked against the same map; it is re-captured when the map's tensors change (after every keyframe).
If a frame outgrows the speculative binning capacity during replay (gsr_forward_status), the caller repeats the frame eagerly."""
import ctypes as C
import math
import os
import types

import torch

from diff_gaussian_rasterization import _C
from diff_gaussian_rasterization import raw as _raw
import gaussian_renderer
from gaussian_renderer import render
import slam_losses

from . import _lib
from .camera import Camera

# GSR_TRACK_STEP=0: the iteration through autograd (render -> weighted_l1_loss -> backward -> pose_step: ten launches), as rounds 2-5 ran it
FUSED_STEP = os.environ.get("GSR_TRACK_STEP", "1") not in ("", "0")


class TrackingGraph:
    def __init__(self, gaussians, pipeline_params, background, config, proto: Camera):
        self.gaussians, self.pipe, self.background, self.config = gaussians, pipeline_params, background, config
        dev = proto.device
        H, W = int(proto.image_height), int(proto.image_width)
        self.cam = Camera(1, None, None, torch.eye(4), proto.projection_matrix, proto.fx, proto.fy, proto.cx, proto.cy, proto.FoVx, proto.FoVy,
                          H, W, 0.0, None, device=dev)
        self.gt_image = torch.zeros((3, H, W), device=dev)
        self.gt_depth = torch.zeros((1, H, W), device=dev)
        self.w_rgb = torch.zeros((1, H, W), device=dev)
        self.w_dep = torch.zeros((1, H, W), device=dev)
        self.alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
        lr = config["Training"]["lr"]
        self.lrs = (lr["cam_rot_delta"], lr["cam_trans_delta"], 0.01)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.static = None
        if bool(gaussians.dygs.any()):
            self.static = gaussians.dygs == False  # noqa: E712
            self.static._gsr_gather = _raw.gather_from_mask(self.static)
        self.graph = None
        self.version = self.model_version(gaussians)
        # straight to the fused rasterizer call when render() would take that route anyway: render()'s extras (a fresh zero tensor for the
        # screen-space gradients, the visibility mask) are two launches per iteration that tracking never looks at
        self.direct = gaussian_renderer._fused_prologue_ok(gaussians, pipeline_params, self.static, False) and gaussians.get_xyz.shape[0] > 0
        self.means2D = torch.zeros_like(gaussians.get_xyz)
        bg_ok = isinstance(background, torch.Tensor) and background.is_cuda and background.dtype == torch.float32 and background.is_contiguous()
        self.fused = FUSED_STEP and self.direct and bg_ok
        # ... and with DETACHED parameters: tracking reads the pose gradient only, so the backward pass runs in its pose-only mode
        # (GSR_BACKWARD_POSE_ONLY: no parameter-gradient stores, no covariance -> scale / rotation chain, nothing to zero per iteration)
        g = gaussians
        self.frozen = types.SimpleNamespace(_xyz=g._xyz.detach(), _scaling=g._scaling.detach(), _rotation=g._rotation.detach(), _opacity=g._opacity.detach(),
                                            _features_dc=g._features_dc.detach(), _features_rest=g._features_rest.detach(), dygs=g.dygs,
                                            active_sh_degree=g.active_sh_degree, get_xyz=g._xyz.detach())

    @staticmethod
    def model_version(g):
        """Changes whenever what a captured graph (or the snapshot an eager tracker took) depends on is replaced: the parameter tensors, the SH
        coefficients of higher degree, the active SH degree, the dynamic-subset mask (its storage and its in-place version)."""
        return (tuple(int(t.data_ptr()) for t in (g._xyz, g._scaling, g._rotation, g._opacity, g._features_dc, g._features_rest, g.dygs))
                + (int(g._xyz.shape[0]), int(g.active_sh_degree), int(g.dygs._version)))

    # ---- per frame -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load(self, viewpoint):
        """Copy the frame into the static slot: start pose, exposure, ground truth and the loss weights of get_loss_tracking
        (utils/slam_utils.py:65-77,118-135; they depend on the frame only)."""
        c = self.cam
        c.update_RT(viewpoint.R, viewpoint.T)
        c.reset_pose_optimizer()
        c.exposure_a.copy_(viewpoint.exposure_a)
        c.exposure_b.copy_(viewpoint.exposure_b)
        c.cam_rot_delta.zero_()
        c.cam_trans_delta.zero_()
        gt_image, gt_depth, _, _, t_rgb, t_dep = slam_losses._keyframe_constants(self.config, viewpoint, c.device)
        w_rgb, w_dep = slam_losses.tracking_loss_weights(self.config, viewpoint, gt_image, gt_depth, rm_dynamic=True, mask=None, base=(t_rgb, t_dep))
        self.gt_image.copy_(gt_image)
        self.gt_depth.copy_(gt_depth.view_as(self.gt_depth))
        self.w_rgb.copy_(w_rgb.view_as(self.w_rgb))
        self.w_dep.copy_(w_dep.view_as(self.w_dep))

    @torch.no_grad()
    def store(self, viewpoint):
        viewpoint.update_RT(self.cam.R, self.cam.T)
        viewpoint.exposure_a.copy_(self.cam.exposure_a)
        viewpoint.exposure_b.copy_(self.cam.exposure_b)

    # ---- the iteration ---------------------------------------------------------------------------------------------------
    def _fused_iteration(self):
        """The whole iteration as ONE C call, seven launches (include/slam_map.h: gsr_track_step): the loss's cotangents in the tile kernel's
        epilogue, pose-only backward, and one tail launch for the gradient sums + the camera step. Pixel cotangents and the pose gradient are
        the bits of the autograd route; the two exposure gradients are summed per tile instead of per 256 strided pixels (equal to rounding)."""
        c, g = self.cam, self.frozen
        dev = c.device
        H, W = int(c.image_height), int(c.image_width)
        P = int(g._xyz.shape[0] if self.static is None else self.static._gsr_gather.shape[0])
        M = 1 + (int(g._features_rest.shape[1]) if g._features_rest.numel() else 0)
        img = torch.empty((_C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        color, depth, opacity = img[:_C.NUM_CHANNELS], img[_C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[_C.NUM_CHANNELS + 1:]
        ints = torch.empty((2, P), dtype=torch.int32, device=dev)
        lib = _lib.lib()
        ws = torch.empty((int(lib.gsr_track_workspace_size(W, H)),), dtype=torch.uint8, device=dev)
        geom, binning, imgbuf = _C._Arena(dev), _C._Arena(dev), _C._Arena(dev)
        keep = []
        f_rest = g._features_rest if g._features_rest.numel() else None
        desc = _raw._describe(g._xyz, g._scaling, g._rotation, g._opacity, g._features_dc, f_rest, None, None, None, None, keep,
                              None if self.static is None else self.static._gsr_gather)
        loss = _lib.TrackLoss()
        loss.gt_image, loss.gt_depth = self.gt_image.data_ptr(), self.gt_depth.data_ptr()
        loss.w_rgb, loss.w_depth = self.w_rgb.data_ptr(), self.w_dep.data_ptr()
        loss.alpha, loss.opacity_depth_threshold, loss.opacity_weights = float(self.alpha), 0.95, 1
        step = c._step_desc(None, tuple(float(x) for x in self.lrs), True, 1e-4, True)
        with torch.cuda.device(dev):
            rc = lib.gsr_track_step(geom.cb, None, binning.cb, None, imgbuf.cb, None, P, int(g.active_sh_degree), M, self.background.data_ptr(), W, H,
                                    C.byref(desc), 1.0, c.projection_matrix.data_ptr(), math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5),
                                    color.data_ptr(), depth.data_ptr(), opacity.data_ptr(), ints[0].data_ptr(), ints[1].data_ptr(),
                                    C.byref(loss), C.byref(step), self.means2D.data_ptr(), ws.data_ptr(), _lib.stream(dev))
        _lib.check(rc, "gsr_track_step")
        self.workspace = ws         # (tests read the cotangents and the gradient sums from it)
        return {"render": color, "radii": ints[0], "depth": depth, "opacity": opacity, "n_touched": ints[1]}

    def iteration(self):
        c = self.cam
        if self.fused:
            return self._fused_iteration()
        if self.direct:
            image, radii, depth, opacity, n_touched = gaussian_renderer._render_fused(c, self.frozen, self.background, 1.0, self.means2D, None, None,
                                                                                      None, self.static, False)
            pkg = {"render": image, "radii": radii, "depth": depth, "opacity": opacity, "n_touched": n_touched}
        else:
            pkg = render(c, self.gaussians, self.pipe, self.background, dynamic=False, mask=self.static)
        # the loss VALUE is never read in this loop: only its gradient is taken
        loss = slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], self.gt_image, self.gt_depth, self.w_rgb, self.w_dep, c.exposure_a,
                                            c.exposure_b, self.alpha, opacity=pkg["opacity"], opacity_depth_threshold=0.95, compute_value=False)
        loss.backward(self._one)                  # (an explicit unit gradient: backward() alone fills a ones_like per call)
        c.pose_step(*self.lrs, latch=True)
        if not self.direct and self.gaussians.optimizer is not None:
            self.gaussians.optimizer.zero_grad(set_to_none=True)
        return pkg

    def capture(self, warmup=3):
        """Warm up eagerly (allocator, lazy-mode capacity), then capture one iteration. The slot's state is restored afterwards, so
        capture() does not move the camera."""
        c = self.cam
        keep = [t.detach().clone() for t in (c._R, c._T, c._adam, c.exposure_a, c.exposure_b)]
        self._lazy_before = _C.set_option("lazy", 1)
        s = torch.cuda.Stream(device=c.device)
        s.wait_stream(torch.cuda.current_stream(c.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.iteration()
        torch.cuda.current_stream(c.device).wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.pkg = self.iteration()          # static outputs: after a replay they hold that iteration's render (depth / opacity of the
                                                 # last tracking iteration feed get_median_depth, utils/slam_frontend.py:461)
        _C.set_option("lazy", self._lazy_before)      # the flag only matters while host code runs: replays never consult it
        with torch.no_grad():
            for dst, src in zip((c._R, c._T, c._adam, c.exposure_a, c.exposure_b), keep):
                dst.copy_(src)
            c.cam_rot_delta.zero_()
            c.cam_trans_delta.zero_()
            c.refresh_matrices()
        return self

    def run(self, iters, check_every=5):
        """Replay up to `iters` iterations; stops at the first convergence poll that succeeds. Returns (iterations run, ok) where
        ok = False means a replayed forward pass outgrew its binning buffer and the frame must be redone eagerly."""
        done = 0
        torch.cuda.current_stream(self.cam.device).synchronize()     # eager work queued before (mapping, evaluation renders) may still bump the counter
        self.overflow0 = _C.forward_status()[0]
        for it in range(iters):
            self.graph.replay()
            done += 1
            if (it + 1) % check_every == 0 and self.cam.converged():
                break
        torch.cuda.current_stream(self.cam.device).synchronize()
        return done, _C.forward_status()[0] == self.overflow0

    def release(self):
        self.graph = None
