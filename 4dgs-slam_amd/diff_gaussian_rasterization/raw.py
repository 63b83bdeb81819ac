"""Fused prologue: rasterize straight from the GaussianModel's raw parameters.

The reference's ``render()`` (gaussian_splatting/gaussian_renderer/__init__.py:108-127,159-174) builds the rasterizer inputs
with a chain of small torch kernels per view --

    means3D   = pc._xyz + scatter(dx -> pc.dygs)
    scales    = exp(pc._scaling) [.repeat(1, 3) if isotropic] + scatter(ds -> pc.dygs)
    rotations = normalize(pc._rotation) + scatter(dr -> pc.dygs)
    opacities = sigmoid(pc._opacity)
    shs       = cat(pc._features_dc, pc._features_rest, dim=1)

(scene/gaussian_model.py:60-68,100-128) and autograd replays as many on the way back. ``rasterize_gaussians_raw`` hands the
raw tensors to ``gsr_forward_raw`` / ``gsr_backward_raw`` (include/gs_rasterizer.h), which apply these maps and their chain
rules inside the preprocess / geometry-backward kernels: same image, same gradients, ~25 kernel launches per view fewer.
``gaussian_renderer.render`` takes this route by itself when the model allows it (see ``_fused_prologue_ok`` there).

The six parameter gradients come back as views of ONE allocation, in the optimizer's parameter order
(xyz, f_dc, f_rest, opacity, scaling, rotation; gaussian_model.py:404-434), so ``mapping_shard.GradBucket`` can all-reduce
them in place.
"""
import ctypes as C

import torch

from . import _C
from .autograd import _pose_grad

_f, _i, _vp = C.c_float, C.c_int, C.c_void_p


class _RawInputs(C.Structure):     # gsr_raw_inputs
    _fields_ = [("xyz", _vp), ("log_scales", _vp), ("scale_dim", _i), ("raw_rotations", _vp), ("logit_opacity", _vp),
                ("features_dc", _vp), ("features_rest", _vp), ("dyn_slot", _vp), ("dx", _vp), ("ds", _vp), ("dr", _vp), ("gather", _vp),
                ("flow_dx2", _vp), ("flow_proj1", _vp), ("flow_proj2", _vp), ("delta_mode", _i), ("delta_stride", _i)]


class _RawGrads(C.Structure):      # gsr_raw_grads
    _fields_ = [("xyz", _vp), ("log_scales", _vp), ("raw_rotations", _vp), ("logit_opacity", _vp), ("features_dc", _vp),
                ("features_rest", _vp), ("dx", _vp), ("ds", _vp), ("dr", _vp), ("dx2", _vp)]


_declared = False


def _lib():
    global _declared
    lib = _C.load_library()
    if not _declared:
        lib.gsr_forward_raw.restype = _i
        lib.gsr_forward_raw.argtypes = [_C._ALLOC_FN, _vp, _C._ALLOC_FN, _vp, _C._ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i,
                                        C.POINTER(_RawInputs), _f, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]
        lib.gsr_backward_raw.restype = _i
        lib.gsr_backward_raw.argtypes = [_i, _i, _i, _i, _vp, _i, _i, C.POINTER(_RawInputs), _f, _vp, _vp, _vp, _vp, _f, _f, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_RawGrads), _vp, _i, _vp]
        _declared = True
    return lib


def _f32(t, name, keep):
    """Device pointer of a contiguous float32 (or int32 for dyn_slot) tensor; None / empty -> NULL."""
    if t is None or t.numel() == 0:
        return None
    _C._require_device(t, name)
    tc = t.contiguous()
    keep.append(tc)
    return tc.data_ptr()


def _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, keep, gather=None):
    for t, name in ((xyz, "_xyz"), (log_scales, "_scaling"), (raw_rot, "_rotation"), (logit_opacity, "_opacity"), (f_dc, "_features_dc")):
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if dyn_slot is not None and dyn_slot.dtype != torch.int32:
        raise RuntimeError("dyn_slot must be int32")
    d = _RawInputs()
    d.xyz, d.log_scales, d.scale_dim = _f32(xyz, "_xyz", keep), _f32(log_scales, "_scaling", keep), int(log_scales.shape[-1])
    d.raw_rotations, d.logit_opacity = _f32(raw_rot, "_rotation", keep), _f32(logit_opacity, "_opacity", keep)
    d.features_dc, d.features_rest = _f32(f_dc, "_features_dc", keep), _f32(f_rest, "_features_rest", keep)
    d.dyn_slot, d.dx, d.ds, d.dr = _f32(dyn_slot, "dyn_slot", keep), _f32(dx, "dx", keep), _f32(ds, "ds", keep), _f32(dr, "dr", keep)
    if gather is not None and gather.dtype != torch.int32:
        raise RuntimeError("gather must be int32")
    d.gather = _f32(gather, "gather", keep)
    return d


def dyn_slot_from_mask(dygs: torch.Tensor) -> torch.Tensor:
    """int32[P]: position of every dynamic Gaussian inside ``x[dygs]`` (the row of dx / ds / dr it receives), -1 elsewhere."""
    m = dygs.to(torch.bool)
    return torch.where(m, torch.cumsum(m.to(torch.int32), 0, dtype=torch.int32) - 1, torch.full_like(m, -1, dtype=torch.int32))


def _acc_params(xyz, f_dc, f_rest, logit_opacity, log_scales, raw_rot):
    """The six model parameters in the optimizer's order if every one of them takes part in fused gradient accumulation
    (autograd._accumulation_targets; f_rest may be absent or empty at SH degree 0), else None."""
    from .autograd import ACCUMULATE_ATTR
    ps = (xyz, f_dc, f_rest, logit_opacity, log_scales, raw_rot)
    for k, p in enumerate(ps):
        if k == 2 and (p is None or p.numel() == 0):
            continue
        if not (isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad and getattr(p, ACCUMULATE_ATTR, False) and p.grad is not None):
            return None
    return ps


def _targets(ps, M):
    """The .grad buffers of _acc_params' tensors at backward time (None: fall back to returning the gradients)."""
    out = []
    for k, p in enumerate(ps):
        if k == 2 and (p is None or p.numel() == 0):
            out.append(torch.empty((0,), dtype=torch.float32, device=ps[0].device))
            continue
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or g.device != p.device:
            return None
        out.append(g)
    return out


class _RasterizeGaussiansRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, means2D, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, theta, rho, rs, gather=None):
        _C._require_device(xyz, "_xyz")
        dev = xyz.device
        ctx.rs = rs
        ctx.pose_shapes = (tuple(theta.shape) if isinstance(theta, torch.Tensor) else None,
                           tuple(rho.shape) if isinstance(rho, torch.Tensor) else None)
        ctx.set_materialize_grads(False)
        ctx.gather = gather
        ctx.acc_params = _acc_params(xyz, f_dc, f_rest, logit_opacity, log_scales, raw_rot)
        ctx.empty = gather is not None and gather.numel() == 0
        if ctx.empty:
            # render()'s mask selected nothing (x[mask] with P = 0): like rasterize_points.cu:85 the kernels are skipped -- zero image,
            # empty radii / n_touched, zero gradients
            H, W = int(rs.image_height), int(rs.image_width)
            ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dx, ds, dr)
            z = lambda c: torch.zeros((c, H, W), dtype=torch.float32, device=dev)
            e = torch.zeros((0,), dtype=torch.int32, device=dev)
            ctx.mark_non_differentiable(e)
            return z(_C.NUM_CHANNELS), e, z(1), z(1), e.clone()
        if _C._glue is not None:      # native host glue (csrc/torch_glue.cpp): same calls, marshalled in C++
            with torch.cuda.device(dev):
                (rc, color, radii, geom_t, bin_t, img_t, depth, opacity, n_touched) = _C._glue.rasterize_gaussians_raw(
                    rs.bg, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, float(rs.scale_modifier),
                    rs.viewmatrix, rs.projmatrix, float(rs.tanfovx), float(rs.tanfovy), int(rs.image_height), int(rs.image_width),
                    int(rs.sh_degree), rs.campos, bool(rs.debug), gather, _C._stream(dev))
            ctx.num_rendered = rc
            ctx.M = 1 + (int(f_rest.shape[1]) if f_rest is not None and f_rest.numel() else 0)
            ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, radii, geom_t, bin_t, img_t)
            ctx.mark_non_differentiable(radii, n_touched)
            return color, radii, depth, opacity, n_touched
        lib = _lib()
        P, H, W = int(xyz.shape[0] if gather is None else gather.shape[0]), int(rs.image_height), int(rs.image_width)
        M = 1 + (int(f_rest.shape[1]) if f_rest is not None and f_rest.numel() else 0)
        img = torch.empty((_C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        color, depth, opacity = img[:_C.NUM_CHANNELS], img[_C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[_C.NUM_CHANNELS + 1:]
        ints = torch.empty((2, P), dtype=torch.int32, device=dev)
        radii, n_touched = ints[0], ints[1]
        geom, binning, imgbuf = _C._Arena(dev), _C._Arena(dev), _C._Arena(dev)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, keep, gather)
        with torch.cuda.device(dev):
            rc = lib.gsr_forward_raw(
                geom.cb, None, binning.cb, None, imgbuf.cb, None, P, int(rs.sh_degree), M, _f32(rs.bg, "bg", keep), W, H,
                C.byref(desc), float(rs.scale_modifier), _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep),
                _f32(rs.campos, "campos", keep), float(rs.tanfovx), float(rs.tanfovy),
                color.data_ptr(), depth.data_ptr(), opacity.data_ptr(), radii.data_ptr(), n_touched.data_ptr(),
                int(bool(rs.debug)), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_forward_raw")
        ctx.num_rendered, ctx.M = rc, M
        ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, radii,
                              geom.tensor, binning.tensor, imgbuf.tensor)
        ctx.mark_non_differentiable(radii, n_touched)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, _g_opacity, _g_touched):
        if ctx.empty:
            xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dx, ds, dr = ctx.saved_tensors
            Z = lambda t: None if t is None else torch.zeros_like(t)
            if ctx.acc_params is not None and _targets(ctx.acc_params, 1) is not None:
                Zp = lambda t: None            # accumulating zero into the parameters' .grad: nothing to do
            else:
                Zp = Z
            th, rh = ctx.pose_shapes
            zp = lambda shp: None if shp is None else torch.zeros(shp, dtype=torch.float32, device=xyz.device)
            return (Zp(xyz), torch.zeros((xyz.shape[0], 3), dtype=torch.float32, device=xyz.device), Zp(log_scales), Zp(raw_rot), Zp(logit_opacity), Zp(f_dc),
                    Zp(f_rest) if f_rest is not None and f_rest.numel() else None, None, Z(dx), Z(ds), Z(dr), zp(th), zp(rh), None, None)
        rs, M = ctx.rs, ctx.M
        (xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, radii, geom, binning, imgbuf) = ctx.saved_tensors
        dev = xyz.device
        P, H, W, S = int(xyz.shape[0]), int(rs.image_height), int(rs.image_width), int(log_scales.shape[-1])
        if g_color is None:
            g_color = _zero_cotangent(3, H, W, dev)
        if g_depth is None:
            g_depth = _zero_cotangent(1, H, W, dev)
        th_shape, rho_shape = ctx.pose_shapes
        gather = ctx.gather
        # fused accumulation into the parameters' .grad buffers (autograd._accumulation_targets): the six gradients are then not returned
        targets = _targets(ctx.acc_params, M) if getattr(ctx, "acc_params", None) is not None else None
        # nobody asked for a parameter or delta gradient (camera tracking renders detached Gaussians): GSR_BACKWARD_POSE_ONLY
        pose_only = not any(ctx.needs_input_grad[k] for k in (0, 2, 3, 4, 5, 6, 8, 9, 10))
        if pose_only:
            targets = None
        if _C._glue is not None:
            with torch.cuda.device(dev):
                (g_xyz, g_fdc, g_frest, g_logit, g_ls, g_rot, g_m2d, g_dx, g_ds, g_dr, tau) = _C._glue.rasterize_gaussians_raw_backward(
                    rs.bg, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, float(rs.scale_modifier),
                    rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, float(rs.tanfovx), float(rs.tanfovy), g_color, g_depth,
                    int(rs.sh_degree), rs.campos, radii, geom, int(ctx.num_rendered), binning, imgbuf, bool(rs.debug), gather, _C._stream(dev),
                    targets or [], pose_only)
            opt = lambda t, src: t if src is not None and src.numel() else None
            if targets is not None or pose_only:
                g_xyz = g_ls = g_rot = g_logit = g_fdc = g_frest = None
            if pose_only:
                g_dx = g_ds = g_dr = None
            return (g_xyz, g_m2d, g_ls, g_rot, g_logit, g_fdc, g_frest if (M > 1 and targets is None and not pose_only) else None, None,
                    opt(g_dx, dx) if g_dx is not None else None, opt(g_ds, ds) if g_ds is not None else None, opt(g_dr, dr) if g_dr is not None else None,
                    _pose_grad(tau[3:], th_shape) if th_shape is not None else None,
                    _pose_grad(tau[:3], rho_shape) if rho_shape is not None else None, None, None)
        lib = _lib()
        g_color = g_color if g_color.dtype == torch.float32 else g_color.to(torch.float32)
        g_depth = g_depth if g_depth.dtype == torch.float32 else g_depth.to(torch.float32)
        # one allocation; parameter order of the optimizer (gaussian_model.py:404-434), then the screen-space gradient
        widths = [3, 3, 3 * (M - 1), 1, S, 4, 3]
        if targets is not None or pose_only:
            widths[:6] = [0] * 6
        # with a mask only the selected rows are written: the rest of the gradients is zero
        flat = (torch.empty if gather is None else torch.zeros)((P * sum(widths) + 6,), dtype=torch.float32, device=dev)
        views, o = [], 0
        for w_ in widths:
            views.append(flat[o:o + P * w_])
            o += P * w_
        if targets is not None:
            views[:6] = [t_.view(-1) for t_ in targets]
        g_m2d = views[6].view(P, 3)
        if not pose_only:
            g_xyz, g_fdc, g_frest = views[0].view(P, 3), views[1].view(P, 1, 3), views[2].view(P, M - 1, 3)
            g_logit, g_ls, g_rot = views[3].view(logit_opacity.shape), views[4].view(P, S), views[5].view(P, 4)
        tau = flat[o:o + 6]
        g_dx, g_ds, g_dr = _zero_grads_like(dx, ds, dr)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, keep, gather)
        out = _RawGrads()
        if not pose_only:
            out.xyz, out.log_scales, out.raw_rotations, out.logit_opacity = g_xyz.data_ptr(), g_ls.data_ptr(), g_rot.data_ptr(), g_logit.data_ptr()
            out.features_dc, out.features_rest = g_fdc.data_ptr(), (g_frest.data_ptr() if M > 1 else None)
            out.dx, out.ds, out.dr = (g_dx.data_ptr() if g_dx is not None else None, g_ds.data_ptr() if g_ds is not None else None,
                                      g_dr.data_ptr() if g_dr is not None else None)
        with torch.cuda.device(dev):
            rc = lib.gsr_backward_raw(
                P if gather is None else int(gather.shape[0]), int(rs.sh_degree), M, int(ctx.num_rendered), _f32(rs.bg, "bg", keep), W, H, C.byref(desc), float(rs.scale_modifier),
                _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep), _f32(rs.projmatrix_raw, "projmatrix_raw", keep),
                _f32(rs.campos, "campos", keep), float(rs.tanfovx), float(rs.tanfovy), radii.data_ptr(),
                geom.data_ptr(), binning.data_ptr(), imgbuf.data_ptr(), _f32(g_color, "dL_dcolor", keep), _f32(g_depth, "dL_ddepth", keep),
                g_m2d.data_ptr(), C.byref(out), tau.data_ptr(), int(bool(rs.debug)) | (2 if targets is not None else 0) | (4 if pose_only else 0),
                _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_backward_raw")
        if targets is not None or pose_only:
            g_xyz = g_ls = g_rot = g_logit = g_fdc = g_frest = None
        if pose_only:
            g_dx = g_ds = g_dr = None
        g_rho = _pose_grad(tau[:3], rho_shape) if rho_shape is not None else None
        g_theta = _pose_grad(tau[3:], th_shape) if th_shape is not None else None
        # inputs: xyz, means2D, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, dx, ds, dr, theta, rho, rs
        return (g_xyz, g_m2d, g_ls, g_rot, g_logit, g_fdc, g_frest if (M > 1 and targets is None and not pose_only) else None, None, g_dx, g_ds, g_dr, g_theta, g_rho, None, None)


_ZERO_COTANGENT = {}


def _zero_cotangent(channels, H, W, device):
    """A read-only zero image standing in for the cotangent of an output the loss did not use (the kernels only read it): one tensor per
    shape and device instead of a zero-fill per backward pass."""
    key = (channels, H, W, device.type, device.index)
    hit = _ZERO_COTANGENT.get(key)
    if hit is None:
        if len(_ZERO_COTANGENT) > 16:
            _ZERO_COTANGENT.clear()
        hit = _ZERO_COTANGENT[key] = torch.zeros((channels, H, W), dtype=torch.float32, device=device)
    return hit


def _zero_grads_like(*tensors):
    """Zero-filled fp32 gradient buffers for the given (optional) tensors out of ONE allocation and ONE fill: the kernels only write the
    rows of visible Gaussians, and four separate zeros_like calls are four launches per backward pass."""
    live = [t for t in tensors if t is not None and t.numel()]
    if not live:
        return [None] * len(tensors)
    flat = torch.zeros((sum(t.numel() for t in live),), dtype=torch.float32, device=live[0].device)
    out, o = [], 0
    for t in tensors:
        if t is None or t.numel() == 0:
            out.append(None)
        else:
            out.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
    return out


def gather_from_mask(mask: torch.Tensor) -> torch.Tensor:
    """int32[Pm]: rows selected by render()'s boolean `mask` (x[mask] order). Synchronises, exactly like x[mask] does."""
    return mask.to(torch.bool).nonzero(as_tuple=False).squeeze(1).to(torch.int32)


def rasterize_gaussians_raw(raster_settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, features_dc, features_rest=None,
                            dyn_slot=None, dx=None, ds=None, dr=None, theta=None, rho=None, gather=None):
    """(color[3,H,W], radii[P], depth[1,H,W], opacity[1,H,W], n_touched[P]) of GaussianRasterizer.forward, from raw model parameters.

    ``dyn_slot`` (int32[P], see dyn_slot_from_mask) is required with dx / ds / dr. ``gather`` (int32[Pm], see gather_from_mask)
    rasterizes only those rows, like render()'s ``mask``: radii / n_touched then have Pm entries and every gradient keeps its
    full shape with zeros in the unselected rows. Empty models must be handled by the caller
    (the reference's render() returns None for them)."""
    if xyz.shape[0] == 0:
        raise RuntimeError("rasterize_gaussians_raw: empty model")
    if (dx is not None or ds is not None or dr is not None) and dyn_slot is None:
        raise RuntimeError("rasterize_gaussians_raw: dx / ds / dr need dyn_slot")
    return _RasterizeGaussiansRaw.apply(xyz, means2D, log_scales, raw_rotations, logit_opacity, features_dc, features_rest, dyn_slot,
                                        dx, ds, dr, theta, rho, raster_settings, gather)


# ---- render_flow(), fused (gaussian_renderer/__init__.py:229-361 of the reference) ---------------------------------------------------
class _RasterizeFlowRaw(torch.autograd.Function):
    """(u, v, mask) flow image of render_flow from raw parameters: the two projections, the NDC difference and the mask channel are
    computed by preprocess_fwd, rasterized like colors_precomp with bg = 0, and the colour's gradient is pushed through both projections
    by geometry_bwd. Differentiable in xyz (geometric path only, :261-262,305), d_xyz1 (both paths), d_xyz2 (colour path), d_scaling1,
    d_rotation1; opacity and the scale / rotation bases are constants (:307,326-334)."""

    @staticmethod
    def forward(ctx, xyz, means2D, log_scales, raw_rot, logit_opacity, dyn_slot, dx1, dx2, ds, dr, proj1, proj2, rs):
        _C._require_device(xyz, "_xyz")
        dev = xyz.device
        lib = _lib()
        P, H, W = int(xyz.shape[0]), int(rs.image_height), int(rs.image_width)
        img = torch.empty((_C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        color, depth, opacity = img[:_C.NUM_CHANNELS], img[_C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[_C.NUM_CHANNELS + 1:]
        ints = torch.empty((2, P), dtype=torch.int32, device=dev)
        radii, n_touched = ints[0], ints[1]
        geom, binning, imgbuf = _C._Arena(dev), _C._Arena(dev), _C._Arena(dev)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, xyz, None, dyn_slot, dx1, ds, dr, keep)
        desc.features_dc = None
        desc.flow_dx2, desc.flow_proj1, desc.flow_proj2 = _f32(dx2, "d_xyz2", keep), _f32(proj1, "proj1", keep), _f32(proj2, "proj2", keep)
        with torch.cuda.device(dev):
            rc = lib.gsr_forward_raw(
                geom.cb, None, binning.cb, None, imgbuf.cb, None, P, 0, 1, _f32(rs.bg, "bg", keep), W, H, C.byref(desc), float(rs.scale_modifier),
                _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep), _f32(rs.campos, "campos", keep),
                float(rs.tanfovx), float(rs.tanfovy), color.data_ptr(), depth.data_ptr(), opacity.data_ptr(), radii.data_ptr(),
                n_touched.data_ptr(), int(bool(rs.debug)), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_forward_raw (flow)")
        ctx.rs, ctx.num_rendered = rs, rc
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, dyn_slot, dx1, dx2, ds, dr, proj1, proj2, radii, geom.tensor, binning.tensor, imgbuf.tensor)
        ctx.mark_non_differentiable(radii, n_touched)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, _g_opacity, _g_touched):
        rs = ctx.rs
        (xyz, log_scales, raw_rot, logit_opacity, dyn_slot, dx1, dx2, ds, dr, proj1, proj2, radii, geom, binning, imgbuf) = ctx.saved_tensors
        dev = xyz.device
        lib = _lib()
        P, H, W, S = int(xyz.shape[0]), int(rs.image_height), int(rs.image_width), int(log_scales.shape[-1])
        g_color = _zero_cotangent(3, H, W, dev) if g_color is None else g_color.to(torch.float32)
        g_depth = _zero_cotangent(1, H, W, dev) if g_depth is None else g_depth.to(torch.float32)
        # xyz | scratch for the constants' gradients the kernel writes anyway (log-scale, rotation, opacity) | means2D | tau
        flat = torch.empty((P * (3 + S + 4 + 1 + 3) + 6,), dtype=torch.float32, device=dev)
        o = 0
        def take(w_):
            nonlocal o
            v = flat[o:o + P * w_]
            o += P * w_
            return v
        g_xyz, g_ls, g_rot, g_logit, g_m2d = take(3).view(P, 3), take(S), take(4), take(1), take(3).view(P, 3)
        tau = flat[o:o + 6]
        g_dx1, g_dx2, g_ds, g_dr = _zero_grads_like(dx1, dx2, ds, dr)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, xyz, None, dyn_slot, dx1, ds, dr, keep)
        desc.features_dc = None
        desc.flow_dx2, desc.flow_proj1, desc.flow_proj2 = _f32(dx2, "d_xyz2", keep), _f32(proj1, "proj1", keep), _f32(proj2, "proj2", keep)
        out = _RawGrads()
        out.xyz, out.log_scales, out.raw_rotations, out.logit_opacity = g_xyz.data_ptr(), g_ls.data_ptr(), g_rot.data_ptr(), g_logit.data_ptr()
        p = lambda t: None if t is None else t.data_ptr()
        out.dx, out.ds, out.dr, out.dx2 = p(g_dx1), p(g_ds), p(g_dr), p(g_dx2)
        with torch.cuda.device(dev):
            rc = lib.gsr_backward_raw(
                P, 0, 1, int(ctx.num_rendered), _f32(rs.bg, "bg", keep), W, H, C.byref(desc), float(rs.scale_modifier),
                _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep), _f32(rs.projmatrix_raw, "projmatrix_raw", keep),
                _f32(rs.campos, "campos", keep), float(rs.tanfovx), float(rs.tanfovy), radii.data_ptr(), geom.data_ptr(), binning.data_ptr(),
                imgbuf.data_ptr(), _f32(g_color, "dL_dcolor", keep), _f32(g_depth, "dL_ddepth", keep), g_m2d.data_ptr(), C.byref(out),
                tau.data_ptr(), int(bool(rs.debug)), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_backward_raw (flow)")
        # inputs: xyz, means2D, log_scales, raw_rot, logit_opacity, dyn_slot, dx1, dx2, ds, dr, proj1, proj2, rs
        return (g_xyz, g_m2d, None, None, None, None, g_dx1, g_dx2, g_ds, g_dr, None, None, None)


def rasterize_flow_raw(raster_settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, dyn_slot, d_xyz1, d_xyz2, d_scaling1, d_rotation1,
                       proj1, proj2):
    """render_flow's rasterizer call from raw parameters (see _RasterizeFlowRaw). raster_settings: camera 1, bg = 0, sh_degree 0."""
    return _RasterizeFlowRaw.apply(xyz, means2D, log_scales, raw_rotations, logit_opacity, dyn_slot, d_xyz1, d_xyz2, d_scaling1, d_rotation1,
                                   proj1, proj2, raster_settings)
