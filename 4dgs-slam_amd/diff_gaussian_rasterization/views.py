"""The views of one mapping iteration through the multi-view entry point (include/gs_rasterizer.h: gsr_forward_views /
gsr_backward_views; csrc/gs_views.h): the same Gaussians rendered from V cameras with ONE launch per pipeline stage, and back-propagated
the same way. Per view the results are those of raw.rasterize_gaussians_raw; the parameter gradients are the sum over the views in
view order (with fused accumulation -- FusedAdam's attached bucket -- exactly what V single-view backward passes leave in the buffers).

    outs = rasterize_views_raw(settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, features_dc, features_rest,
                               dyn_slot=..., deltas=[(dx, ds, dr) | None per view], poses=[(theta, rho) per view])
    outs[v] = (color[3,H,W], radii[P], depth[1,H,W], opacity[1,H,W], n_touched[P])

settings: one GaussianRasterizationSettings per view (same image size, field of view, background, SH degree and scale modifier).
means2D: one zeros [P,3] tensor per view whose .grad receives that view's screen-space gradient."""
import ctypes as C
import os

import torch

from . import _C
from .autograd import _pose_grad
from .raw import _RawGrads, _RawInputs, _acc_params, _describe, _f32, _targets, _zero_cotangent, _zero_grads_like

MAX_VIEWS = 12
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float


class _View(C.Structure):       # gsr_view
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("projmatrix_raw", _vp), ("cam_pos", _vp), ("dx", _vp), ("ds", _vp), ("dr", _vp),
                ("out_color", _vp), ("out_depth", _vp), ("out_opacity", _vp), ("radii", _vp), ("n_touched", _vp),
                ("geometry_user", _vp), ("binning_user", _vp), ("image_user", _vp),
                ("geom_buffer", _vp), ("binning_buffer", _vp), ("image_buffer", _vp), ("num_rendered", _i),
                ("dL_dcolor", _vp), ("dL_ddepth", _vp), ("dL_dmean2D", _vp), ("ddx", _vp), ("dds", _vp), ("ddr", _vp), ("dL_dtau_sum", _vp),
                ("flow_dx2", _vp), ("flow_proj1", _vp), ("flow_proj2", _vp), ("ddx2", _vp), ("flow_clip", _vp)]


_declared = False
_arenas = {}            # id -> tensor holder of the allocations of the call in flight (the C callbacks name them by id)


def _alloc(user, nbytes):
    h = _arenas[int(user)]
    h["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=h["dev"])
    return h["t"].data_ptr()


_alloc_cb = _C._ALLOC_FN(_alloc)


def _lib():
    global _declared
    lib = _C.load_library()
    if not _declared:
        lib.gsr_forward_views.restype = _i
        lib.gsr_forward_views.argtypes = [_i, C.POINTER(_View), _C._ALLOC_FN, _C._ALLOC_FN, _C._ALLOC_FN, _i, _i, _i, _vp, _i, _i,
                                          C.POINTER(_RawInputs), _f, _f, _f, _i, _vp]
        lib.gsr_views_scratch_size.restype = C.c_size_t
        lib.gsr_views_scratch_size.argtypes = [_i, _i, _i, _i]
        lib.gsr_backward_views.restype = _i
        lib.gsr_backward_views.argtypes = [_i, C.POINTER(_View), _i, _i, _i, _vp, _i, _i, C.POINTER(_RawInputs), _f, _f, _f,
                                           C.POINTER(_RawGrads), _vp, _i, _vp]
        _declared = True
    return lib


_NATIVE_MARSHALLING = os.environ.get("GSR_NATIVE_VIEWS", "1") != "0"


def _glue():
    """The native glue's marshalling of the two calls (csrc/torch_glue.cpp rasterize_views_forward / _backward), or None: the ctypes code
    below does the same job in Python (~0.3 ms of host time per call at 6-10 views) and remains the reference path."""
    g = _C._glue
    return g if (_NATIVE_MARSHALLING and g is not None and hasattr(g, "rasterize_views_forward")) else None


def _camera_lists(settings):
    return ([rs.viewmatrix for rs in settings], [rs.projmatrix for rs in settings], [rs.projmatrix_raw for rs in settings], [rs.campos for rs in settings])


class _RasterizeViewsRaw(torch.autograd.Function):
    """inputs: xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, settings (list), then six per view:
    means2D, dx, ds, dr, theta, rho. outputs: five per view: color, radii, depth, opacity, n_touched."""

    @staticmethod
    def forward(ctx, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, settings, *per_view):
        _C._require_device(xyz, "_xyz")
        lib = _lib()
        dev, V = xyz.device, len(settings)
        rs0 = settings[0]
        P, H, W = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width)
        M = 1 + (int(f_rest.shape[1]) if f_rest is not None and f_rest.numel() else 0)
        ctx.settings, ctx.V, ctx.M = settings, V, M
        ctx.set_materialize_grads(False)
        ctx.acc_params = _acc_params(xyz, f_dc, f_rest, logit_opacity, log_scales, raw_rot)
        ctx.pose_shapes = [(tuple(per_view[6 * v + 4].shape) if isinstance(per_view[6 * v + 4], torch.Tensor) else None,
                            tuple(per_view[6 * v + 5].shape) if isinstance(per_view[6 * v + 5], torch.Tensor) else None) for v in range(V)]
        glue = _glue()
        if glue is not None:
            cams = _camera_lists(settings)
            with torch.cuda.device(dev):
                img, ints, rendered, state = glue.rasterize_views_forward(
                    rs0.bg, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest if (f_rest is not None and f_rest.numel()) else None, dyn_slot, *cams,
                    [per_view[6 * v + 1] for v in range(V)], [per_view[6 * v + 2] for v in range(V)], [per_view[6 * v + 3] for v in range(V)], [], [], [],
                    float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), H, W, int(rs0.sh_degree), bool(rs0.debug), _C._stream(dev), [])
            ctx.num_rendered = [int(r) for r in rendered]
            deltas = [per_view[6 * v + k] for v in range(V) for k in (1, 2, 3)]
            ctx.n_state = len(state)
            ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, ints, *state, *deltas)
            outs = []
            for v in range(V):
                outs += [img[v, :_C.NUM_CHANNELS], ints[v, 0], img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[v, _C.NUM_CHANNELS + 1:], ints[v, 1]]
            ctx.mark_non_differentiable(*[outs[5 * v + k] for v in range(V) for k in (1, 4)])      # every view's radii and n_touched, in ONE call (a call replaces the set)
            return tuple(outs)
        img = torch.empty((V, _C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        ints = torch.empty((V, 2, P), dtype=torch.int32, device=dev)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, None, None, None, keep)
        views = (_View * V)()
        base = id(ctx) & 0x3FFFFFFFFFFF
        holders = []
        for v in range(V):
            rs, w = settings[v], views[v]
            _, dx, ds, dr = per_view[6 * v: 6 * v + 4]
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            w.dx, w.ds, w.dr = _f32(dx, "dx", keep), _f32(ds, "ds", keep), _f32(dr, "dr", keep)
            w.out_color, w.out_depth = img[v, :_C.NUM_CHANNELS].data_ptr(), img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1].data_ptr()
            w.out_opacity, w.radii, w.n_touched = img[v, _C.NUM_CHANNELS + 1:].data_ptr(), ints[v, 0].data_ptr(), ints[v, 1].data_ptr()
            hs = [{"dev": dev, "t": None} for _ in range(3)]
            holders.append(hs)
            for k, h in enumerate(hs):
                _arenas[base + 3 * v + k] = h
            w.geometry_user, w.binning_user, w.image_user = base + 3 * v, base + 3 * v + 1, base + 3 * v + 2
        try:
            with torch.cuda.device(dev):
                rc = lib.gsr_forward_views(V, views, _alloc_cb, _alloc_cb, _alloc_cb, P, int(rs0.sh_degree), M, _f32(rs0.bg, "bg", keep), W, H,
                                           C.byref(desc), float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), int(bool(rs0.debug)),
                                           _C._stream(dev))
        finally:
            for v in range(V):
                for k in range(3):
                    _arenas.pop(base + 3 * v + k, None)
        if rc < 0:
            _C._err(lib, rc, "gsr_forward_views")
        ctx.num_rendered = [int(views[v].num_rendered) for v in range(V)]
        state = [holders[v][k]["t"] for v in range(V) for k in range(3)]          # geometry, binning, image per view
        deltas = [per_view[6 * v + k] for v in range(V) for k in (1, 2, 3)]
        ctx.n_state = len(state)
        ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, ints, *state, *deltas)
        outs = []
        for v in range(V):
            outs += [img[v, :_C.NUM_CHANNELS], ints[v, 0], img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[v, _C.NUM_CHANNELS + 1:], ints[v, 1]]
        ctx.mark_non_differentiable(*[outs[5 * v + k] for v in range(V) for k in (1, 4)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib()
        V, M, settings = ctx.V, ctx.M, ctx.settings
        saved = ctx.saved_tensors
        xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, ints = saved[:8]
        state = saved[8:8 + ctx.n_state]
        deltas = saved[8 + ctx.n_state:]
        dev = xyz.device
        rs0 = settings[0]
        P, H, W, S = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width), int(log_scales.shape[-1])
        targets = _targets(ctx.acc_params, M) if ctx.acc_params is not None else None
        param_idx = (0, 1, 2, 3, 4, 5)
        delta_needed = any(ctx.needs_input_grad[8 + 6 * v + k] for v in range(V) for k in (1, 2, 3))
        pose_only = not any(ctx.needs_input_grad[k] for k in param_idx) and not delta_needed
        if pose_only:
            targets = None
        glue = _glue()
        if glue is not None:
            cot = lambda g_, c: _zero_cotangent(c, H, W, dev) if g_ is None else g_
            with torch.cuda.device(dev):
                pg, per_view_out, dl = glue.rasterize_views_backward(
                    rs0.bg, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest if (f_rest is not None and f_rest.numel()) else None, dyn_slot,
                    *_camera_lists(settings), [deltas[3 * v] for v in range(V)], [deltas[3 * v + 1] for v in range(V)], [deltas[3 * v + 2] for v in range(V)],
                    [], [], [], float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), H, W, int(rs0.sh_degree), ints, list(state),
                    ctx.num_rendered, [cot(grads[5 * v], 3) for v in range(V)], [cot(grads[5 * v + 2], 1) for v in range(V)],
                    [t_.view(-1) for t_ in targets] if targets is not None else [], targets is not None, pose_only, bool(rs0.debug), _C._stream(dev))
            if targets is not None or pose_only:
                res = [None] * 8
            else:
                res = [pg[0].view(P, 3), pg[4].view(P, S), pg[5].view(P, 4), pg[3].view(logit_opacity.shape), pg[1].view(P, 1, 3),
                       pg[2].view(P, M - 1, 3) if M > 1 else None, None, None]
            for v in range(V):
                th_shape, rho_shape = ctx.pose_shapes[v]
                tau = per_view_out[v, P * 3:]
                res += [per_view_out[v, :P * 3].view(P, 3), dl[4 * v], dl[4 * v + 2], dl[4 * v + 3],
                        _pose_grad(tau[3:], th_shape) if th_shape is not None else None, _pose_grad(tau[:3], rho_shape) if rho_shape is not None else None]
            return tuple(res)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, dyn_slot, None, None, None, keep)
        widths = [3, 3, 3 * (M - 1), 1, S, 4]
        if targets is not None or pose_only:
            own = None
            gviews = [t_.view(-1) for t_ in targets] if targets is not None else [None] * 6
        else:
            own = torch.empty((P * sum(widths),), dtype=torch.float32, device=dev)
            gviews, o = [], 0
            for w_ in widths:
                gviews.append(own[o:o + P * w_])
                o += P * w_
        out = _RawGrads()
        if not pose_only:
            out.xyz, out.features_dc, out.features_rest = gviews[0].data_ptr(), gviews[1].data_ptr(), (gviews[2].data_ptr() if M > 1 else None)
            out.logit_opacity, out.log_scales, out.raw_rotations = gviews[3].data_ptr(), gviews[4].data_ptr(), gviews[5].data_ptr()
        per_view_out = torch.empty((V, P * 3 + 6), dtype=torch.float32, device=dev)      # screen-space gradient + pose sum per view
        views = (_View * V)()
        delta_grads = []
        zero_deltas = [None] * (3 * V) if pose_only else _zero_grads_like(*deltas[:3 * V])      # one allocation, one fill for all views
        for v in range(V):
            rs, w = settings[v], views[v]
            g_color, g_depth = grads[5 * v], grads[5 * v + 2]
            if g_color is None:
                g_color = _zero_cotangent(3, H, W, dev)
            if g_depth is None:
                g_depth = _zero_cotangent(1, H, W, dev)
            dx, ds, dr = deltas[3 * v: 3 * v + 3]
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            w.dx, w.ds, w.dr = _f32(dx, "dx", keep), _f32(ds, "ds", keep), _f32(dr, "dr", keep)
            w.radii = ints[v, 0].data_ptr()
            w.geom_buffer, w.binning_buffer, w.image_buffer = state[3 * v].data_ptr(), state[3 * v + 1].data_ptr(), state[3 * v + 2].data_ptr()
            w.num_rendered = ctx.num_rendered[v]
            w.dL_dcolor, w.dL_ddepth = _f32(g_color.to(torch.float32), "dL_dcolor", keep), _f32(g_depth.to(torch.float32), "dL_ddepth", keep)
            w.dL_dmean2D, w.dL_dtau_sum = per_view_out[v, :P * 3].data_ptr(), per_view_out[v, P * 3:].data_ptr()
            gd = tuple(zero_deltas[3 * v: 3 * v + 3])
            delta_grads.append(gd)
            w.ddx, w.dds, w.ddr = (None if g is None else g.data_ptr() for g in gd)
        scratch = None
        if not pose_only:
            scratch = torch.empty((int(lib.gsr_views_scratch_size(V, P, M, S)),), dtype=torch.uint8, device=dev)
        flags = int(bool(rs0.debug)) | (2 if targets is not None else 0) | (4 if pose_only else 0)
        with torch.cuda.device(dev):
            rc = lib.gsr_backward_views(V, views, P, int(rs0.sh_degree), M, _f32(rs0.bg, "bg", keep), W, H, C.byref(desc), float(rs0.scale_modifier),
                                        float(rs0.tanfovx), float(rs0.tanfovy), C.byref(out), None if scratch is None else scratch.data_ptr(), flags,
                                        _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_backward_views")
        if own is not None:
            g_xyz, g_fdc, g_frest = gviews[0].view(P, 3), gviews[1].view(P, 1, 3), (gviews[2].view(P, M - 1, 3) if M > 1 else None)
            g_logit, g_ls, g_rot = gviews[3].view(logit_opacity.shape), gviews[4].view(P, S), gviews[5].view(P, 4)
        else:
            g_xyz = g_fdc = g_frest = g_logit = g_ls = g_rot = None
        res = [g_xyz, g_ls, g_rot, g_logit, g_fdc, g_frest, None, None]
        for v in range(V):
            th_shape, rho_shape = ctx.pose_shapes[v]
            tau = per_view_out[v, P * 3:]
            gdx, gds, gdr = delta_grads[v]
            res += [per_view_out[v, :P * 3].view(P, 3), gdx, gds, gdr,
                    _pose_grad(tau[3:], th_shape) if th_shape is not None else None, _pose_grad(tau[:3], rho_shape) if rho_shape is not None else None]
        return tuple(res)


class _RasterizeViewsNet(torch.autograd.Function):
    """The V keyframes of one mapping iteration of render(dynamic=True) (gaussian_renderer/__init__.py:149-157): the same Gaussians, every
    view moved by its own output of the 4DGaussians deformation network. inputs: xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest,
    net_out [V, P, 10] = (dx | ds | dr) per view (deformation.deform_network.forward_views), settings (list), then three per view: means2D,
    theta, rho. outputs: five per view, as _RasterizeViewsRaw. The deltas are added IN FRONT of the activations inside the kernels
    (gsr_raw_inputs.delta_mode = 1, delta_stride = 10: no [P, 3] / [P, 4] copies of the network's output, and its gradient comes back as
    one [V, P, 10] tensor that the fused MLP's backward reads as it is)."""

    @staticmethod
    def forward(ctx, xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, net_out, settings, *per_view):
        _C._require_device(xyz, "_xyz")
        lib = _lib()
        dev, V = xyz.device, len(settings)
        rs0 = settings[0]
        P, H, W = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width)
        if net_out.dtype != torch.float32 or tuple(net_out.shape) != (V, P, 10):
            raise RuntimeError(f"net_out must be float32 [{V}, {P}, 10], got {net_out.dtype} {tuple(net_out.shape)}")
        _C._require_device(net_out, "net_out")
        net_out = net_out.contiguous()
        M = 1 + (int(f_rest.shape[1]) if f_rest is not None and f_rest.numel() else 0)
        ctx.settings, ctx.V, ctx.M = settings, V, M
        ctx.set_materialize_grads(False)
        ctx.acc_params = _acc_params(xyz, f_dc, f_rest, logit_opacity, log_scales, raw_rot)
        ctx.pose_shapes = [(tuple(per_view[3 * v + 1].shape) if isinstance(per_view[3 * v + 1], torch.Tensor) else None,
                            tuple(per_view[3 * v + 2].shape) if isinstance(per_view[3 * v + 2], torch.Tensor) else None) for v in range(V)]
        img = torch.empty((V, _C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        ints = torch.empty((V, 2, P), dtype=torch.int32, device=dev)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, None, None, None, None, keep)
        desc.delta_mode, desc.delta_stride = 1, 10
        views = (_View * V)()
        base = id(ctx) & 0x3FFFFFFFFFFF
        holders = []
        net_ptr = net_out.data_ptr()
        for v in range(V):
            rs, w = settings[v], views[v]
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            row0 = net_ptr + 4 * 10 * P * v
            w.dx, w.ds, w.dr = row0, row0 + 12, row0 + 24
            w.out_color, w.out_depth = img[v, :_C.NUM_CHANNELS].data_ptr(), img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1].data_ptr()
            w.out_opacity, w.radii, w.n_touched = img[v, _C.NUM_CHANNELS + 1:].data_ptr(), ints[v, 0].data_ptr(), ints[v, 1].data_ptr()
            hs = [{"dev": dev, "t": None} for _ in range(3)]
            holders.append(hs)
            for k, h in enumerate(hs):
                _arenas[base + 3 * v + k] = h
            w.geometry_user, w.binning_user, w.image_user = base + 3 * v, base + 3 * v + 1, base + 3 * v + 2
        try:
            with torch.cuda.device(dev):
                rc = lib.gsr_forward_views(V, views, _alloc_cb, _alloc_cb, _alloc_cb, P, int(rs0.sh_degree), M, _f32(rs0.bg, "bg", keep), W, H,
                                           C.byref(desc), float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), int(bool(rs0.debug)),
                                           _C._stream(dev))
        finally:
            for v in range(V):
                for k in range(3):
                    _arenas.pop(base + 3 * v + k, None)
        if rc < 0:
            _C._err(lib, rc, "gsr_forward_views (network deltas)")
        ctx.num_rendered = [int(views[v].num_rendered) for v in range(V)]
        state = [holders[v][k]["t"] for v in range(V) for k in range(3)]
        ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, net_out, ints, *state)
        outs = []
        for v in range(V):
            outs += [img[v, :_C.NUM_CHANNELS], ints[v, 0], img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[v, _C.NUM_CHANNELS + 1:], ints[v, 1]]
        ctx.mark_non_differentiable(*[outs[5 * v + k] for v in range(V) for k in (1, 4)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib()
        V, M, settings = ctx.V, ctx.M, ctx.settings
        saved = ctx.saved_tensors
        xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, net_out, ints = saved[:8]
        state = saved[8:]
        dev = xyz.device
        rs0 = settings[0]
        P, H, W, S = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width), int(log_scales.shape[-1])
        targets = _targets(ctx.acc_params, M) if ctx.acc_params is not None else None
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, f_dc, f_rest, None, None, None, None, keep)
        desc.delta_mode, desc.delta_stride = 1, 10
        widths = [3, 3, 3 * (M - 1), 1, S, 4]
        if targets is not None:
            own = None
            gviews = [t_.view(-1) for t_ in targets]
        else:
            own = torch.empty((P * sum(widths),), dtype=torch.float32, device=dev)
            gviews, o = [], 0
            for w_ in widths:
                gviews.append(own[o:o + P * w_])
                o += P * w_
        out = _RawGrads()
        out.xyz, out.features_dc, out.features_rest = gviews[0].data_ptr(), gviews[1].data_ptr(), (gviews[2].data_ptr() if M > 1 else None)
        out.logit_opacity, out.log_scales, out.raw_rotations = gviews[3].data_ptr(), gviews[4].data_ptr(), gviews[5].data_ptr()
        per_view_out = torch.empty((V, P * 3 + 6), dtype=torch.float32, device=dev)      # screen-space gradient + pose sum per view
        g_net = torch.empty_like(net_out)             # every row is written: geometry_bwd stores zeros for the Gaussians a view does not see
        views = (_View * V)()
        net_ptr, gnet_ptr = net_out.data_ptr(), g_net.data_ptr()
        for v in range(V):
            rs, w = settings[v], views[v]
            g_color, g_depth = grads[5 * v], grads[5 * v + 2]
            if g_color is None:
                g_color = _zero_cotangent(3, H, W, dev)
            if g_depth is None:
                g_depth = _zero_cotangent(1, H, W, dev)
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            row0, grow0 = net_ptr + 4 * 10 * P * v, gnet_ptr + 4 * 10 * P * v
            w.dx, w.ds, w.dr = row0, row0 + 12, row0 + 24
            w.ddx, w.dds, w.ddr = grow0, grow0 + 12, grow0 + 24
            w.radii = ints[v, 0].data_ptr()
            w.geom_buffer, w.binning_buffer, w.image_buffer = state[3 * v].data_ptr(), state[3 * v + 1].data_ptr(), state[3 * v + 2].data_ptr()
            w.num_rendered = ctx.num_rendered[v]
            w.dL_dcolor, w.dL_ddepth = _f32(g_color.to(torch.float32), "dL_dcolor", keep), _f32(g_depth.to(torch.float32), "dL_ddepth", keep)
            w.dL_dmean2D, w.dL_dtau_sum = per_view_out[v, :P * 3].data_ptr(), per_view_out[v, P * 3:].data_ptr()
        scratch = torch.empty((int(lib.gsr_views_scratch_size(V, P, M, S)),), dtype=torch.uint8, device=dev)
        flags = int(bool(rs0.debug)) | (2 if targets is not None else 0)
        with torch.cuda.device(dev):
            rc = lib.gsr_backward_views(V, views, P, int(rs0.sh_degree), M, _f32(rs0.bg, "bg", keep), W, H, C.byref(desc), float(rs0.scale_modifier),
                                        float(rs0.tanfovx), float(rs0.tanfovy), C.byref(out), scratch.data_ptr(), flags, _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_backward_views (network deltas)")
        if os.environ.get("GSR_DEBUG_ZERO_ROWS"):            # development: how many (view, Gaussian) pairs receive no gradient at all
            print("zero rows of the network cotangent:", float((g_net.abs().amax(dim=-1) == 0).float().mean()), flush=True)
        if own is not None:
            res = [gviews[0].view(P, 3), gviews[4].view(P, S), gviews[5].view(P, 4), gviews[3].view(logit_opacity.shape), gviews[1].view(P, 1, 3),
                   gviews[2].view(P, M - 1, 3) if M > 1 else None, g_net, None]
        else:
            res = [None, None, None, None, None, None, g_net, None]
        for v in range(V):
            th_shape, rho_shape = ctx.pose_shapes[v]
            tau = per_view_out[v, P * 3:]
            res += [per_view_out[v, :P * 3].view(P, 3), _pose_grad(tau[3:], th_shape) if th_shape is not None else None,
                    _pose_grad(tau[:3], rho_shape) if rho_shape is not None else None]
        return tuple(res)


def rasterize_views_net(settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, features_dc, features_rest, net_out, poses=None):
    """render(dynamic=True) of V cameras at once: net_out [V, P, 10] is the deformation network's (dx | ds | dr) per camera
    (deform_network.forward_views); the result per camera is that of raw.rasterize_gaussians_raw on (xyz + dx, log_scales + ds,
    raw_rotations + dr), see _RasterizeViewsNet."""
    V = len(settings)
    if xyz.shape[0] == 0:
        raise RuntimeError("rasterize_views_net: empty model")
    if not views_supported(settings):
        raise RuntimeError("rasterize_views_net: the views must share image size, field of view, background, SH degree and scale modifier")
    poses = poses or [(None, None)] * V
    flat = []
    for v in range(V):
        flat += [means2D[v], poses[v][0], poses[v][1]]
    outs = _RasterizeViewsNet.apply(xyz, log_scales, raw_rotations, logit_opacity, features_dc, features_rest, net_out, list(settings), *flat)
    return [tuple(outs[5 * v: 5 * v + 5]) for v in range(V)]


class _RasterizeFlowViewsRaw(torch.autograd.Function):
    """render_flow (raw.rasterize_flow_raw) for several (camera 1, camera 2) pairs of one mapping iteration at once.
    inputs: xyz, log_scales, raw_rot, logit_opacity, dyn_slot, settings (list, camera 1 of every pair), then seven per view:
    means2D, d_xyz1, d_xyz2, d_scaling1, d_rotation1, proj1, proj2. outputs: five per view: color (u, v, mask), radii, depth, opacity,
    n_touched. Differentiable like the single call: xyz (geometric path; summed over the views in view order), d_xyz1, d_xyz2,
    d_scaling1, d_rotation1 per view."""

    @staticmethod
    def forward(ctx, xyz, log_scales, raw_rot, logit_opacity, dyn_slot, settings, clips, *per_view):
        _C._require_device(xyz, "_xyz")
        lib = _lib()
        dev, V = xyz.device, len(settings)
        rs0 = settings[0]
        P, H, W = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width)
        ctx.settings, ctx.V = settings, V
        clips = list(clips) if clips is not None else [None] * V      # per view: int32 [4] device tensor (gsr_view.flow_clip) or None
        for c in clips:
            if c is not None and not (isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.int32 and c.numel() == 4 and c.is_contiguous()):
                raise ValueError("flow clips: int32 [4] contiguous device tensors (or None)")
        ctx.clips = clips                                             # the kernels of a captured call keep reading these addresses
        ctx.set_materialize_grads(False)
        glue = _glue()
        if glue is not None:
            col = lambda k: [per_view[7 * v + k] for v in range(V)]
            with torch.cuda.device(dev):
                img, ints, rendered, state = glue.rasterize_views_forward(
                    rs0.bg, xyz, log_scales, raw_rot, logit_opacity, None, None, dyn_slot, *_camera_lists(settings), col(1), col(3), col(4), col(2), col(5), col(6),
                    float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), H, W, 0, bool(rs0.debug), _C._stream(dev), clips)
            ctx.num_rendered = [int(r) for r in rendered]
            ctx.n_state = len(state)
            ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, dyn_slot, ints, *state, *[per_view[7 * v + k] for v in range(V) for k in range(1, 7)])
            outs = []
            for v in range(V):
                outs += [img[v, :_C.NUM_CHANNELS], ints[v, 0], img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[v, _C.NUM_CHANNELS + 1:], ints[v, 1]]
            ctx.mark_non_differentiable(*[outs[5 * v + k] for v in range(V) for k in (1, 4)])      # every view's radii and n_touched, in ONE call (a call replaces the set)
            return tuple(outs)
        img = torch.empty((V, _C.NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
        ints = torch.empty((V, 2, P), dtype=torch.int32, device=dev)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, xyz, None, dyn_slot, None, None, None, keep)
        desc.features_dc = None
        views = (_View * V)()
        base = id(ctx) & 0x3FFFFFFFFFFF
        holders = []
        for v in range(V):
            rs, w = settings[v], views[v]
            _, dx1, dx2, ds, dr, proj1, proj2 = per_view[7 * v: 7 * v + 7]
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            w.dx, w.ds, w.dr = _f32(dx1, "d_xyz1", keep), _f32(ds, "d_scaling1", keep), _f32(dr, "d_rotation1", keep)
            w.flow_dx2, w.flow_proj1, w.flow_proj2 = _f32(dx2, "d_xyz2", keep), _f32(proj1, "proj1", keep), _f32(proj2, "proj2", keep)
            w.flow_clip = clips[v].data_ptr() if clips[v] is not None else None
            w.out_color, w.out_depth = img[v, :_C.NUM_CHANNELS].data_ptr(), img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1].data_ptr()
            w.out_opacity, w.radii, w.n_touched = img[v, _C.NUM_CHANNELS + 1:].data_ptr(), ints[v, 0].data_ptr(), ints[v, 1].data_ptr()
            hs = [{"dev": dev, "t": None} for _ in range(3)]
            holders.append(hs)
            for k, h in enumerate(hs):
                _arenas[base + 3 * v + k] = h
            w.geometry_user, w.binning_user, w.image_user = base + 3 * v, base + 3 * v + 1, base + 3 * v + 2
        try:
            with torch.cuda.device(dev):
                rc = lib.gsr_forward_views(V, views, _alloc_cb, _alloc_cb, _alloc_cb, P, 0, 1, _f32(rs0.bg, "bg", keep), W, H, C.byref(desc),
                                           float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), int(bool(rs0.debug)), _C._stream(dev))
        finally:
            for v in range(V):
                for k in range(3):
                    _arenas.pop(base + 3 * v + k, None)
        if rc < 0:
            _C._err(lib, rc, "gsr_forward_views (flow)")
        ctx.num_rendered = [int(views[v].num_rendered) for v in range(V)]
        state = [holders[v][k]["t"] for v in range(V) for k in range(3)]
        ctx.n_state = len(state)
        ctx.save_for_backward(xyz, log_scales, raw_rot, logit_opacity, dyn_slot, ints, *state, *[per_view[7 * v + k] for v in range(V) for k in range(1, 7)])
        outs = []
        for v in range(V):
            outs += [img[v, :_C.NUM_CHANNELS], ints[v, 0], img[v, _C.NUM_CHANNELS:_C.NUM_CHANNELS + 1], img[v, _C.NUM_CHANNELS + 1:], ints[v, 1]]
        ctx.mark_non_differentiable(*[outs[5 * v + k] for v in range(V) for k in (1, 4)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib()
        V, settings = ctx.V, ctx.settings
        saved = ctx.saved_tensors
        xyz, log_scales, raw_rot, logit_opacity, dyn_slot, ints = saved[:6]
        state = saved[6:6 + ctx.n_state]
        rest = saved[6 + ctx.n_state:]                      # per view: dx1, dx2, ds, dr, proj1, proj2
        dev = xyz.device
        rs0 = settings[0]
        P, H, W, S = int(xyz.shape[0]), int(rs0.image_height), int(rs0.image_width), int(log_scales.shape[-1])
        glue = _glue()
        if glue is not None:
            col = lambda k: [rest[6 * v + k] for v in range(V)]          # per view: dx1, dx2, ds, dr, proj1, proj2
            cot = lambda g_, c: _zero_cotangent(c, H, W, dev) if g_ is None else g_
            with torch.cuda.device(dev):
                pg, per_view_out, dl = glue.rasterize_views_backward(
                    rs0.bg, xyz, log_scales, raw_rot, logit_opacity, None, None, dyn_slot, *_camera_lists(settings), col(0), col(2), col(3), col(1), col(4), col(5),
                    float(rs0.scale_modifier), float(rs0.tanfovx), float(rs0.tanfovy), H, W, 0, ints, list(state), ctx.num_rendered,
                    [cot(grads[5 * v], 3) for v in range(V)], [cot(grads[5 * v + 2], 1) for v in range(V)], [], False, False, bool(rs0.debug), _C._stream(dev))
            res = [pg[0], None, None, None, None, None, None]
            for v in range(V):
                res += [per_view_out[v, :P * 3].view(P, 3), dl[4 * v], dl[4 * v + 1], dl[4 * v + 2], dl[4 * v + 3], None, None]
            return tuple(res)
        keep = []
        desc = _describe(xyz, log_scales, raw_rot, logit_opacity, xyz, None, dyn_slot, None, None, None, keep)
        desc.features_dc = None
        g_xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
        out = _RawGrads()
        out.xyz = g_xyz.data_ptr()
        per_view_out = torch.empty((V, P * 3 + 6), dtype=torch.float32, device=dev)      # screen-space gradient + (unused) pose sum per view
        zero = _zero_grads_like(*[rest[6 * v + k] for v in range(V) for k in range(4)])   # one fill for every view's delta gradients
        views = (_View * V)()
        for v in range(V):
            rs, w = settings[v], views[v]
            g_color, g_depth = grads[5 * v], grads[5 * v + 2]
            g_color = _zero_cotangent(3, H, W, dev) if g_color is None else g_color
            g_depth = _zero_cotangent(1, H, W, dev) if g_depth is None else g_depth
            dx1, dx2, ds, dr, proj1, proj2 = rest[6 * v: 6 * v + 6]
            w.viewmatrix, w.projmatrix = _f32(rs.viewmatrix, "viewmatrix", keep), _f32(rs.projmatrix, "projmatrix", keep)
            w.projmatrix_raw, w.cam_pos = _f32(rs.projmatrix_raw, "projmatrix_raw", keep), _f32(rs.campos, "campos", keep)
            w.dx, w.ds, w.dr = _f32(dx1, "d_xyz1", keep), _f32(ds, "d_scaling1", keep), _f32(dr, "d_rotation1", keep)
            w.flow_dx2, w.flow_proj1, w.flow_proj2 = _f32(dx2, "d_xyz2", keep), _f32(proj1, "proj1", keep), _f32(proj2, "proj2", keep)
            w.radii = ints[v, 0].data_ptr()
            w.geom_buffer, w.binning_buffer, w.image_buffer = state[3 * v].data_ptr(), state[3 * v + 1].data_ptr(), state[3 * v + 2].data_ptr()
            w.num_rendered = ctx.num_rendered[v]
            w.dL_dcolor, w.dL_ddepth = _f32(g_color.to(torch.float32), "dL_dcolor", keep), _f32(g_depth.to(torch.float32), "dL_ddepth", keep)
            w.dL_dmean2D, w.dL_dtau_sum = per_view_out[v, :P * 3].data_ptr(), per_view_out[v, P * 3:].data_ptr()
            gd1, gd2, gds, gdr = zero[4 * v: 4 * v + 4]
            w.ddx, w.ddx2, w.dds, w.ddr = (None if g is None else g.data_ptr() for g in (gd1, gd2, gds, gdr))
        scratch = torch.empty((int(lib.gsr_views_scratch_size(V, P, 1, S)),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gsr_backward_views(V, views, P, 0, 1, _f32(rs0.bg, "bg", keep), W, H, C.byref(desc), float(rs0.scale_modifier),
                                        float(rs0.tanfovx), float(rs0.tanfovy), C.byref(out), scratch.data_ptr(), int(bool(rs0.debug)), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_backward_views (flow)")
        res = [g_xyz, None, None, None, None, None, None]
        for v in range(V):
            gd1, gd2, gds, gdr = zero[4 * v: 4 * v + 4]
            res += [per_view_out[v, :P * 3].view(P, 3), gd1, gd2, gds, gdr, None, None]
        return tuple(res)


def rasterize_flow_views_raw(settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, dyn_slot, flows, clips=None):
    """flows[v] = (d_xyz1, d_xyz2, d_scaling1, d_rotation1, proj1, proj2) of the v-th (camera 1 -> camera 2) pair; settings[v] describes
    camera 1 (bg = 0, sh_degree 0). clips[v]: None or an int32 [4] device tensor, the tile rectangle [x0, y0, x1, y1) the caller reads of that
    image (gsr_view.flow_clip). Returns the tuples raw.rasterize_flow_raw returns, one per pair."""
    V = len(settings)
    if xyz.shape[0] == 0:
        raise RuntimeError("rasterize_flow_views_raw: empty model")
    if not views_supported(settings) or dyn_slot is None:
        raise RuntimeError("rasterize_flow_views_raw: the views must share image size, field of view and background, and need dyn_slot")
    flat = []
    for v in range(V):
        dx1, dx2, ds, dr, proj1, proj2 = flows[v]
        flat += [means2D[v], dx1, dx2, ds, dr, proj1, proj2]
    outs = _RasterizeFlowViewsRaw.apply(xyz, log_scales, raw_rotations, logit_opacity, dyn_slot, list(settings), clips, *flat)
    return [tuple(outs[5 * v: 5 * v + 5]) for v in range(V)]


def views_supported(settings):
    """Same image, field of view, background TENSOR (the same object / storage), SH degree, scale modifier and debug flag for every view,
    and no more than MAX_VIEWS."""
    if not (1 <= len(settings) <= MAX_VIEWS):
        return False
    a = settings[0]
    return all(int(s.image_height) == int(a.image_height) and int(s.image_width) == int(a.image_width) and float(s.tanfovx) == float(a.tanfovx)
               and float(s.tanfovy) == float(a.tanfovy) and int(s.sh_degree) == int(a.sh_degree) and float(s.scale_modifier) == float(a.scale_modifier)
               and bool(s.debug) == bool(a.debug) and (s.bg is a.bg or s.bg.data_ptr() == a.bg.data_ptr()) for s in settings)   # (no value comparison:
    # torch.equal would synchronise with the device on every call; callers pass the one background tensor of the system)


def rasterize_views_raw(settings, xyz, means2D, log_scales, raw_rotations, logit_opacity, features_dc, features_rest=None, dyn_slot=None,
                        deltas=None, poses=None):
    V = len(settings)
    if xyz.shape[0] == 0:
        raise RuntimeError("rasterize_views_raw: empty model")
    if not views_supported(settings):
        raise RuntimeError("rasterize_views_raw: the views must share image size, field of view, background, SH degree and scale modifier")
    deltas = deltas or [None] * V
    poses = poses or [(None, None)] * V
    flat = []
    for v in range(V):
        dx, ds, dr = deltas[v] if deltas[v] is not None else (None, None, None)
        if (dx is not None or ds is not None or dr is not None) and dyn_slot is None:
            raise RuntimeError("rasterize_views_raw: dx / ds / dr need dyn_slot")
        flat += [means2D[v], dx, ds, dr, poses[v][0], poses[v][1]]
    outs = _RasterizeViewsRaw.apply(xyz, log_scales, raw_rotations, logit_opacity, features_dc, features_rest, dyn_slot, list(settings), *flat)
    return [tuple(outs[5 * v: 5 * v + 5]) for v in range(V)]
