"""Autograd boundary of the MI355X rasterizer: the counterpart of ``_RasterizeGaussians``
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:48-171).

``means2D``, ``theta`` and ``rho`` do not influence the forward value; they only receive gradients
(reference :102-104,152-169). Only the colour and depth cotangents are consumed in backward; those of
opacity / radii / n_touched are ignored exactly as the reference does (:108,116-138).
"""
import math

import torch

from . import _C


def _snapshot(args):
    """CPU copies of every tensor argument, taken before a debug-mode call (reference :17-19,90-97)."""
    return tuple(a.cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call(fn, args, debug, dump_path, message):
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print(message)
        raise


def _lean_backward(*args, accumulate_into=None):
    """Backward without the intermediate per-Gaussian gradients nobody reads here (see _C.rasterize_gaussians_backward_fused)."""
    return _C.rasterize_gaussians_backward_fused(*args, lean=True, accumulate_into=accumulate_into)


ACCUMULATE_ATTR = "_gsr_accumulate_grad"


def _accumulation_targets(params):
    """Fused gradient accumulation (opt-in: mapping_shard.GradBucket.attach marks the parameters): when all five Gaussian parameter
    inputs are leaf tensors whose ``.grad`` buffers exist and are marked, the backward kernels ADD this view's gradients to those
    buffers and the Function returns None for them -- autograd's AccumulateGrad (a read-modify-write of five tensors per view) and the
    zero rows of invisible Gaussians disappear. Same values as autograd's own accumulation, bit for bit (one fp32 add per element and
    view, in view order). Not for torch.autograd.grad(): that call wants the gradients returned."""
    out = []
    for p in params:
        g = getattr(p, "grad", None)
        if not (isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad and getattr(p, ACCUMULATE_ATTR, False) and g is not None
                and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g.device == p.device):
            return None
        out.append(g)
    return out


def _pose_grad(vec3, shape):
    """A pose gradient (3 values) shaped like the 3-element input it belongs to, else [1,3] as the reference returns it."""
    if shape is not None and len(shape) >= 1 and math.prod(shape) == 3:
        return vec3.view(shape)
    return vec3.view(1, 3)


def _camera_block(rs):
    """(scale_modifier, cov3D slot filled by caller, viewmatrix, projmatrix, projmatrix_raw, tanfovx, tanfovy)."""
    return rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, rs.tanfovx, rs.tanfovy


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings):
        rs = raster_settings
        # positional order of _C.rasterize_gaussians == rasterize_points.h:18-39
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                *_camera_block(rs), rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        (num_rendered, color, radii, geom_buf, binning_buf, img_buf, depth, opacity, n_touched) = _call(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.pose_shapes = (tuple(theta.shape) if isinstance(theta, torch.Tensor) else None,
                           tuple(rho.shape) if isinstance(rho, torch.Tensor) else None)
        ctx.set_materialize_grads(False)   # unused cotangents (opacity, radii, n_touched) arrive as None, not as zero-filled tensors
        ctx.acc_params = None
        if colors_precomp.numel() == 0 and cov3Ds_precomp.numel() == 0 and _accumulation_targets((means3D, sh, opacities, scales, rotations)):
            ctx.acc_params = (means3D, sh, opacities, scales, rotations)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf, img_buf)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_radii, grad_out_depth, grad_out_opacity, grad_n_touched):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf, img_buf = ctx.saved_tensors
        hw = (rs.image_height, rs.image_width)
        if grad_out_color is None:
            grad_out_color = means3D.new_zeros((3, *hw))
        if grad_out_depth is None:
            grad_out_depth = means3D.new_zeros((1, *hw))
        # positional order of _C.rasterize_gaussians_backward == rasterize_points.h:41-65
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                *_camera_block(rs), grad_out_color, grad_out_depth, sh, rs.sh_degree, rs.campos,
                geom_buf, ctx.num_rendered, binning_buf, img_buf, rs.debug)
        targets = _accumulation_targets(ctx.acc_params) if ctx.acc_params is not None else None
        bw = _lean_backward if targets is None else (lambda *a_: _lean_backward(*a_, accumulate_into=targets))
        (g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rot, g_tau, tau) = _call(
            bw, args, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        if targets is not None:      # already added to the parameters' .grad by the kernels
            g_means3D = g_sh = g_opacity = g_scales = g_rot = None
        # gradients of inputs that were not given (empty tensors in, empty tensors out) are None for autograd
        g_colors = g_colors if g_colors.numel() else None
        g_cov3D = g_cov3D if g_cov3D.numel() else None
        # per-Gaussian [rho | theta] rows -> one pose gradient, each returned as [1,3] (reference :152-154: torch.sum over [P,6]);
        # the sum is produced by the backward kernels themselves (tau = float32[6])
        # The reference returns them as [1,3] and lets autograd sum_to_size them onto the (3,) camera parameters, which costs
        # one reduction kernel each; a 3-element input gets its gradient in its own shape instead (same values).
        th_shape, rho_shape = ctx.pose_shapes
        g_rho, g_theta = _pose_grad(tau[:3], rho_shape), _pose_grad(tau[3:], th_shape)
        # one gradient per forward input, in input order (reference :157-169)
        return (g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rot, g_cov3D, g_theta, g_rho, None)
