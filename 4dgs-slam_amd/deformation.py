"""The 4D-Gaussians deformation network on the MI355X library -- the counterpart of the reference's utils/deformation.py with
the same public names (deform_network, Deformation, poc_fre, initialize_weights), constructor arguments, parameter / buffer
names (a reference state dict loads unchanged) and return values:  (means3D, scales, rotations, dx, ds, dr).

What differs is where the work happens: the HexPlane field -- 24 grid_sample calls, 20 products and a concat per call in the
reference (gaussian_splatting/utils/hexplane.py:81-112) -- is one fused HIP launch per direction (hexplane.py ->
include/deformation_field.h); the dense layers keep library GEMMs for the forward and input-gradient products and use the
library's split-K MFMA kernel for the weight gradients (PointwiseLinear below).  The sin/cos positional embeddings the
reference computes and then discards (deform_network.forward_dynamic builds [n, 63] / [n, 15] / [n, 20] embeddings of which only
the leading raw columns are read, utils/deformation.py:198-213,78,113,122,132) are not computed."""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from diff_gaussian_rasterization import _C
from hexplane import HexPlaneField

FUSED_MLP = os.environ.get("GSR_FUSED_MLP", "1") != "0"   # fused forward / backward kernels for the shipped MLP structure


class _Mlp(ctypes.Structure):
    _fields_ = [("W0", ctypes.c_void_p), ("b0", ctypes.c_void_p), ("W1", ctypes.c_void_p * 3), ("b1", ctypes.c_void_p * 3),
                ("W2", ctypes.c_void_p * 3), ("b2", ctypes.c_void_p * 3), ("in_dim", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib_cache = None


def _lib():
    global _lib_cache
    if _lib_cache is None:
        lib = _C.load_library()
        i64, vp, i = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        lib.gsr_linear_wgrad_workspace_size.restype = ctypes.c_size_t
        lib.gsr_linear_wgrad_workspace_size.argtypes = [i64, i, i]
        lib.gsr_linear_wgrad.restype = i
        lib.gsr_linear_wgrad.argtypes = [i64, i, i, vp, i64, vp, i64, vp, vp, vp, vp]
        lib.gsr_deform_mlp_forward.restype = i
        lib.gsr_deform_mlp_forward.argtypes = [ctypes.POINTER(_Mlp), i64, vp, vp, vp]
        lib.gsr_deform_mlp_backward.restype = i
        lib.gsr_deform_mlp_backward.argtypes = [ctypes.POINTER(_Mlp), i64, vp, vp, vp, vp, vp, vp]
        lib.gsr_deform_mlp_grad_count.restype = ctypes.c_size_t
        lib.gsr_deform_mlp_grad_count.argtypes = [i]
        lib.gsr_deform_mlp_workspace_size.restype = ctypes.c_size_t
        lib.gsr_deform_mlp_workspace_size.argtypes = [i]
        lib.gsr_deform_mlp_backward_rows.restype = i
        lib.gsr_deform_mlp_backward_rows.argtypes = [ctypes.POINTER(_Mlp), i64, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.gsr_row_mask_workspace_size.restype = ctypes.c_size_t
        lib.gsr_row_mask_workspace_size.argtypes = [i, i64]
        lib.gsr_row_mask.restype = i
        lib.gsr_row_mask.argtypes = [i, i64, i, vp, vp, vp, vp, vp, vp]
        _lib_cache = lib
    return _lib_cache


class _PointwiseLinear(torch.autograd.Function):
    """y = x Wᵀ + b over a long batch of points.  Forward and the input gradient are library GEMMs; the WEIGHT gradient --
    a 64x128-or-smaller output reduced over every point, the shape the vendor GEMM handles worst (3.4 of the 4.5 ms the MLP took
    forward+backward at 200k points) -- is the split-K MFMA kernel of include/deformation_field.h (gsr_linear_wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if x.stride(1) != 1:
                x = x.contiguous()
            if gy.stride(1) != 1 or gy.stride(0) < gy.shape[1]:      # e.g. an expanded cotangent of sum()
                gy = gy.contiguous()
            n, in_dim, out_dim = x.shape[0], x.shape[1], gy.shape[1]
            lib = _lib()
            gw = torch.empty((out_dim, in_dim), dtype=torch.float32, device=x.device)
            gb = torch.empty((out_dim,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
            ws = torch.empty((lib.gsr_linear_wgrad_workspace_size(n, in_dim, out_dim),), dtype=torch.uint8, device=x.device)
            with torch.cuda.device(x.device):
                rc = lib.gsr_linear_wgrad(n, in_dim, out_dim, x.data_ptr(), x.stride(0), gy.data_ptr(), gy.stride(0), gw.data_ptr(),
                                          gb.data_ptr() if gb is not None else None, ws.data_ptr(), _C._stream(x.device))
            if rc < 0:
                _C._err(lib, rc, "gsr_linear_wgrad")
        return gx, gw, gb


class _FusedDeformMLP(torch.autograd.Function):
    """out [n, 10] = (dx, ds, dr) of the shipped deformation MLP from the HexPlane features, one fused kernel per direction
    (csrc/gs_mlp.h) plus seven weight-gradient launches.  params: W0, b0, then (W1, b1, W2, b2) for the position, scale and
    rotation heads -- the nn.Linear tensors themselves."""

    @staticmethod
    def _describe(in_dim, params):
        m = _Mlp()
        m.W0, m.b0, m.in_dim = params[0].data_ptr(), params[1].data_ptr(), in_dim
        for j in range(3):
            W1, b1, W2, b2 = params[2 + 4 * j:6 + 4 * j]
            m.W1[j], m.b1[j], m.W2[j], m.b2[j] = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
        return m

    @staticmethod
    def forward(ctx, feat, *params):
        feat = feat.contiguous()
        params = tuple(p.detach().contiguous() for p in params)
        glue = _C._glue
        if glue is not None and hasattr(glue, "deform_mlp_forward"):      # native host glue (csrc/torch_glue.cpp)
            out = glue.deform_mlp_forward(feat.detach(), list(params), _C._stream(feat.device))
            ctx.save_for_backward(feat, *params)
            return out
        n, in_dim = feat.shape
        dev = feat.device
        out = torch.empty((n, 10), dtype=torch.float32, device=dev)
        lib = _lib()
        m = _FusedDeformMLP._describe(in_dim, params)
        with torch.cuda.device(dev):
            rc = lib.gsr_deform_mlp_forward(ctypes.byref(m), n, feat.data_ptr(), out.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_deform_mlp_forward")
        ctx.save_for_backward(feat, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, *params = ctx.saved_tensors
        glue = _C._glue
        if glue is not None and hasattr(glue, "deform_mlp_backward"):
            res = glue.deform_mlp_backward(feat, list(params), dout, _C._stream(feat.device))
            return (res[0] if ctx.needs_input_grad[0] else None, *res[1:])
        n, in_dim = feat.shape
        dev = feat.device
        dout = dout.contiguous()
        lib = _lib()
        dfeat = torch.empty((n, in_dim), dtype=torch.float32, device=dev)
        flat = torch.empty((lib.gsr_deform_mlp_grad_count(in_dim),), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.gsr_deform_mlp_workspace_size(in_dim),), dtype=torch.uint8, device=dev)
        m = _FusedDeformMLP._describe(in_dim, params)
        with torch.cuda.device(dev):
            rc = lib.gsr_deform_mlp_backward(ctypes.byref(m), n, feat.data_ptr(), dout.data_ptr(), dfeat.data_ptr(), flat.data_ptr(), ws.data_ptr(),
                                             _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_deform_mlp_backward")
        grads, off = [], 0                                       # the flat layout of include/deformation_field.h is the parameter order
        for p in params:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (dfeat if ctx.needs_input_grad[0] else None, *grads)


class _DeformViews(torch.autograd.Function):
    """out [V, n, 10] = (dx | ds | dr) of the shipped deformation network for the SAME points at the V times of one mapping iteration's keyframes
    (gaussian_renderer/__init__.py:112,149-157 call it once per keyframe): ONE autograd node for the HexPlane field of all views
    (hexplane._HexPlaneFeaturesViews' kernels) and the MLP over all V * n rows.

    Backward: most (view, Gaussian) rows of the cotangent are exactly zero -- the Gaussian is outside that view's frustum or behind saturated
    pixels (63 % of the 8 x 500k rows at BASELINE config #3) -- and a zero row contributes nothing to any gradient. gsr_row_mask lists the
    others; the MLP's backward runs over the list only (gsr_deform_mlp_backward_rows) and the field's backward skips the unlisted rows by
    their view bit (gsr_hexplane_backward_views' view_mask); nothing is synchronised with the host. GSR_ROW_MASK=0 processes every row.

    inputs: xyz [n, >= 3], times (tuple of floats), aabb, n_levels, then the 6 * n_levels planes, then the 14 MLP tensors (W0, b0, and
    W1, b1, W2, b2 of the position, scale and rotation heads)."""

    @staticmethod
    def forward(ctx, xyz, times, aabb, n_levels, *tensors):
        import hexplane as hp
        planes, params = tensors[:6 * n_levels], tuple(p.detach().contiguous() for p in tensors[6 * n_levels:])
        _C._require_device(xyz, "pts")
        if xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.shape[1] < 3:
            raise ValueError(f"the deformation network expects fp32 points [n, 3], got {xyz.dtype} {tuple(xyz.shape)}")
        if xyz.stride(1) != 1:
            xyz = xyz.contiguous()
        V, n, dev = len(times), xyz.shape[0], xyz.device
        levels = [[p.detach() for p in planes[6 * l:6 * l + 6]] for l in range(n_levels)]
        in_dim = n_levels * levels[0][0].shape[1]
        feat = torch.empty((V, n, in_dim), dtype=torch.float32, device=dev)
        out = torch.empty((V, n, 10), dtype=torch.float32, device=dev)
        field = hp._describe(levels, aabb)
        tv = (ctypes.c_float * V)(*[float(t) for t in times])
        hl, lib = hp._lib(), _lib()
        m = _FusedDeformMLP._describe(in_dim, params)
        with torch.cuda.device(dev):
            rc = hl.gsr_hexplane_forward_views(ctypes.byref(field), n, xyz.data_ptr(), xyz.stride(0), V, tv, feat.data_ptr(), _C._stream(dev))
            if rc < 0:
                _C._err(hl, rc, "gsr_hexplane_forward_views")
            rc = lib.gsr_deform_mlp_forward(ctypes.byref(m), V * n, feat.data_ptr(), out.data_ptr(), _C._stream(dev))
            if rc < 0:
                _C._err(lib, rc, "gsr_deform_mlp_forward")
        ctx.save_for_backward(xyz, aabb if aabb is not None else torch.empty(0), feat, *planes, *params)
        ctx.n_levels, ctx.has_aabb, ctx.times = n_levels, aabb is not None, tv
        return out

    @staticmethod
    def backward(ctx, dout):
        import hexplane as hp
        xyz, aabb, feat, *rest = ctx.saved_tensors
        n_levels = ctx.n_levels
        planes, params = rest[:6 * n_levels], rest[6 * n_levels:]
        aabb = aabb if ctx.has_aabb else None
        V, n, in_dim = feat.shape
        dev = feat.device
        dout = dout.contiguous()
        hl, lib = hp._lib(), _lib()
        stream = _C._stream(dev)
        m = _FusedDeformMLP._describe(in_dim, params)
        dfeat = torch.empty((V, n, in_dim), dtype=torch.float32, device=dev)         # rows that are not listed stay unwritten and are never read
        flat_g = torch.empty((lib.gsr_deform_mlp_grad_count(in_dim),), dtype=torch.float32, device=dev)
        ws_mlp = torch.empty((lib.gsr_deform_mlp_workspace_size(in_dim),), dtype=torch.uint8, device=dev)
        use_mask = os.environ.get("GSR_ROW_MASK", "1") != "0" and V <= 32 and V * n < 2 ** 31
        mask = rows = None
        with torch.cuda.device(dev):
            if use_mask:
                mask = torch.empty((n,), dtype=torch.int32, device=dev)
                rows = torch.empty((V * n + 1,), dtype=torch.int32, device=dev)       # the list, then its length
                ws_rows = torch.empty((lib.gsr_row_mask_workspace_size(V, n),), dtype=torch.uint8, device=dev)
                rc = lib.gsr_row_mask(V, n, 10, dout.data_ptr(), mask.data_ptr(), rows.data_ptr(), rows[V * n:].data_ptr(), ws_rows.data_ptr(), stream)
                if rc < 0:
                    _C._err(lib, rc, "gsr_row_mask")
            rc = lib.gsr_deform_mlp_backward_rows(ctypes.byref(m), V * n, feat.data_ptr(), dout.data_ptr(), dfeat.data_ptr(), flat_g.data_ptr(), ws_mlp.data_ptr(),
                                                  rows.data_ptr() if use_mask else None, rows[V * n:].data_ptr() if use_mask else None, stream)
            if rc < 0:
                _C._err(lib, rc, "gsr_deform_mlp_backward_rows")
        mlp_grads, off = [], 0
        for p in params:
            mlp_grads.append(flat_g[off:off + p.numel()].view(p.shape))
            off += p.numel()
        # ---- the field, all views ----
        levels = [[p.detach() for p in planes[6 * l:6 * l + 6]] for l in range(n_levels)]
        need_plane = list(ctx.needs_input_grad[4:4 + 6 * n_levels])
        sizes = [p.numel() if need else 0 for p, need in zip(planes, need_plane)]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)               # one fill for every plane gradient
        views, o = [], 0
        for p, need, sz in zip(planes, need_plane, sizes):
            views.append(torch.as_strided(flat, p.shape, p.stride(), o) if need else None)
            o += sz
        gxyz = torch.empty((n, 3), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        field = hp._describe(levels, aabb, [views[6 * l:6 * l + 6] for l in range(n_levels)])
        size = hl.gsr_hexplane_backward_views_workspace_size(ctypes.byref(field), n, V)
        if size == 0:
            raise RuntimeError("deform_network.forward_views: plane geometry not covered by the batched field backward")
        ws = torch.empty(size, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = hl.gsr_hexplane_backward_views(ctypes.byref(field), n, xyz.data_ptr(), xyz.stride(0), V, ctx.times, dfeat.data_ptr(),
                                                mask.data_ptr() if use_mask else None, gxyz.data_ptr() if gxyz is not None else None, ws.data_ptr(), stream)
        if rc < 0:
            _C._err(hl, rc, "gsr_hexplane_backward_views")
        if gxyz is not None and xyz.shape[1] > 3:
            full = torch.zeros_like(xyz)
            full[:, :3] = gxyz
            gxyz = full
        return (gxyz, None, None, None, *views, *mlp_grads)


class PointwiseLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys, same value) whose weight gradient runs on the MI355X library when the
    input is a 2-D fp32 batch of points on a HIP device with at most 128 features; anything else takes nn.Linear's own path."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 128 and x.shape[0] > 0 and torch.is_grad_enabled():
            return _PointwiseLinear.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def poc_fre(input_data, poc_buf):
    """utils/deformation.py:227-233."""
    poc_buf = poc_buf.to(device=input_data.device)
    emb = (input_data.unsqueeze(-1) * poc_buf).flatten(-2)
    return torch.cat([input_data, emb.sin(), emb.cos()], -1)


def initialize_weights(m):
    """utils/deformation.py:220-226: Xavier-uniform weights; biases keep nn.Linear's default."""
    if isinstance(m, nn.Linear):
        init.xavier_uniform_(m.weight, gain=1)


def _head(width, out_dim):
    return nn.Sequential(nn.ReLU(), PointwiseLinear(width, width), nn.ReLU(), PointwiseLinear(width, out_dim))


class Deformation(nn.Module):
    """utils/deformation.py:17-165."""

    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=[], args=None, device="cuda"):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_time, self.skips, self.grid_pe = D, W, input_ch, input_ch_time, skips, grid_pe
        self.no_grid = args.no_grid
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires)
        self.args = args
        if args.empty_voxel:
            raise NotImplementedError("empty_voxel (a DenseGrid occupancy mask; marked 'useless' and off in the reference's "
                                      "arguments/__init__.py:101) is not part of the MI355X deformation field")
        if args.static_mlp:
            self.static_mlp = _head(W, 1)
        self.ratio = 0
        self.device = device
        self.create_net()

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_max, xyz_min)

    def create_net(self):
        grid_out_dim = self.grid.feat_dim * 3 if self.grid_pe != 0 else self.grid.feat_dim
        layers = [PointwiseLinear(4 if self.no_grid else grid_out_dim, self.W)]
        for _ in range(self.D - 1):
            layers += [nn.ReLU(), PointwiseLinear(self.W, self.W)]
        self.feature_out = nn.Sequential(*layers).to(self.device)
        self.pos_deform = _head(self.W, 3).to(self.device)
        self.scales_deform = _head(self.W, 3).to(self.device)
        self.rotations_deform = _head(self.W, 4).to(self.device)
        self.opacity_deform = _head(self.W, 1).to(self.device)
        self.shs_deform = _head(self.W, 16 * 3).to(self.device)

    def query_time(self, rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb):
        time_emb = time_emb.to(device=rays_pts_emb.device)
        if self.no_grid:
            h = torch.cat([rays_pts_emb[:, :3], time_emb[:, :1]], -1)
        else:
            h = self.grid(rays_pts_emb[:, :3], time_emb[:, :1])
            if self.grid_pe > 1:
                h = poc_fre(h, self.grid_pe)
        return self.feature_out(h)

    @property
    def get_empty_ratio(self):
        return self.ratio

    def forward(self, rays_pts_emb, scales_emb=None, rotations_emb=None, opacity=None, shs_emb=None, time_feature=None, time_emb=None):
        if time_emb is None:
            return self.forward_static(rays_pts_emb[:, :3])
        return self.forward_dynamic(rays_pts_emb, scales_emb, rotations_emb, opacity, shs_emb, time_feature, time_emb)

    def forward_static(self, rays_pts_emb):
        return rays_pts_emb[:, :3] + self.static_mlp(self.grid(rays_pts_emb[:, :3]))

    def _fused_mlp_ok(self, x):
        """the shipped structure: one trunk layer of width 64, the three delta heads, nothing else in the way"""
        a = self.args
        return (FUSED_MLP and x.is_cuda and x.dtype == torch.float32 and self.D == 1 and self.W == 64
                and not (self.no_grid or a.static_mlp or a.no_dx or a.no_ds or a.no_dr or a.apply_rotation) and self.grid_pe == 0
                and self.grid.feat_dim % 16 == 0 and self.grid.feat_dim <= 128)

    def forward_views(self, points, times):
        """(dx | ds | dr) [V, n, 10] of forward_dynamic for the SAME points at the V times of one mapping iteration's keyframes
        (gaussian_renderer/__init__.py:112,149-157 call the network once per keyframe with `time.repeat(n, 1)`): the field through
        HexPlaneField.forward_views, the MLP once over all V * n rows. None when the shipped structure does not apply (caller: one
        forward_dynamic per view). The rasterizer adds the deltas to the raw parameters itself (gsr_raw_inputs.delta_mode = 1)."""
        if not self._fused_mlp_ok(points) or points.shape[0] == 0:
            return None
        import hexplane as hp
        planes = hp._PlaneList(self.grid.grids)
        if not hp.views_supported(planes, len(times)):
            return None
        heads = (self.pos_deform, self.scales_deform, self.rotations_deform)
        params = [self.feature_out[0].weight, self.feature_out[0].bias]
        for h in heads:
            params += [h[1].weight, h[1].bias, h[3].weight, h[3].bias]
        return _DeformViews.apply(points[:, :3], tuple(float(t) for t in times), self.grid.aabb, planes.n_levels, *planes.flat, *params)

    def forward_dynamic(self, rays_pts_emb, scales_emb, rotations_emb, opacity_emb, shs_emb, time_feature, time_emb):
        if self._fused_mlp_ok(rays_pts_emb):
            feat = self.grid(rays_pts_emb[:, :3], time_emb.to(device=rays_pts_emb.device)[:, :1])
            if feat.shape[0] > 0:
                heads = (self.pos_deform, self.scales_deform, self.rotations_deform)
                params = [self.feature_out[0].weight, self.feature_out[0].bias]
                for h in heads:
                    params += [h[1].weight, h[1].bias, h[3].weight, h[3].bias]
                out = _FusedDeformMLP.apply(feat, *params)
                dx, ds, dr = out[:, 0:3], out[:, 3:6], out[:, 6:10]
                return rays_pts_emb[:, :3] + dx, scales_emb[:, :3] + ds, rotations_emb[:, :4] + dr, dx, ds, dr
        hidden = self.query_time(rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb)
        a = self.args
        mask = self.static_mlp(hidden) if a.static_mlp else None       # None = the reference's all-ones mask (:107)
        keep = (lambda x: x) if mask is None else (lambda x: x * mask)
        dx = ds = dr = None                                             # the reference leaves these unbound when disabled
        if a.no_dx:
            pts = rays_pts_emb[:, :3]
        else:
            dx = self.pos_deform(hidden)
            pts = keep(rays_pts_emb[:, :3]) + dx
        if a.no_ds:
            scales = scales_emb[:, :3]
        else:
            ds = self.scales_deform(hidden)
            scales = keep(scales_emb[:, :3]) + ds
        if a.no_dr:
            rotations = rotations_emb[:, :4]
        else:
            dr = self.rotations_deform(hidden)
            if a.apply_rotation:
                rotations = batch_quaternion_multiply(rotations_emb, dr)
            else:
                rotations = rotations_emb[:, :4] + dr
        # no_do / no_dshs: the opacity and SH heads' results never leave the reference's forward (:134-149); not evaluated
        return pts, scales, rotations, dx, ds, dr

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


def batch_quaternion_multiply(q1, q2):
    """gaussian_splatting/utils/graphics_utils.py:134-157: row-wise Hamilton product of (w, x, y, z) quaternions, normalised."""
    a, b = q1[:, :1], q1[:, 1:4]
    c, d = q2[:, :1], q2[:, 1:4]
    q = torch.cat((a * c - (b * d).sum(dim=1, keepdim=True), a * d + c * b + torch.cross(b, d, dim=1)), dim=1)
    return q / torch.norm(q, dim=1, keepdim=True)


def default_hidden_params(**overrides):
    """The fields of the reference's ModelHiddenParams that the deformation network reads, with the shipped defaults
    (arguments/__init__.py:76-104), as a namespace; keyword arguments override single fields."""
    import types
    a = types.SimpleNamespace(net_width=64, timebase_pe=4, defor_depth=1, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2,
                              timenet_width=64, timenet_output=32, bounds=1.6,
                              kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                              "resolution": [64, 64, 64, 25]},
                              multires=[1, 2, 4, 8], no_dx=False, no_grid=False, no_ds=False, no_dr=False, no_do=True, no_dshs=True,
                              empty_voxel=False, grid_pe=0, static_mlp=False, apply_rotation=False)
    for k, v in overrides.items():
        setattr(a, k, v)
    return a


class deform_network(nn.Module):
    """utils/deformation.py:166-219."""

    def __init__(self, args, device):
        super().__init__()
        times_ch = 2 * args.timebase_pe + 1
        self.timenet = nn.Sequential(nn.Linear(times_ch, args.timenet_width), nn.ReLU(), nn.Linear(args.timenet_width, args.timenet_output))
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, input_ch=3 + 3 * args.posebase_pe * 2, grid_pe=args.grid_pe,
                                           input_ch_time=args.timenet_output, args=args, device=device)
        self.register_buffer("time_poc", torch.FloatTensor([2 ** i for i in range(args.timebase_pe)]))
        self.register_buffer("pos_poc", torch.FloatTensor([2 ** i for i in range(args.posebase_pe)]))
        self.register_buffer("rotation_scaling_poc", torch.FloatTensor([2 ** i for i in range(args.scale_rotation_pe)]))
        self.register_buffer("opacity_poc", torch.FloatTensor([2 ** i for i in range(args.opacity_pe)]))
        self.apply(initialize_weights)

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        return self.forward_dynamic(point, scales, rotations, opacity, shs, times_sel)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.get_empty_ratio

    def forward_static(self, points):
        return self.deformation_net(points)

    def forward_dynamic(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        return self.deformation_net(point, scales, rotations, opacity, shs, None, times_sel)

    def forward_views(self, point, times):
        """[V, n, 10] = (dx | ds | dr) of forward_dynamic(point, ..., times_sel = times[v]) for every v, or None (see Deformation.forward_views)."""
        return self.deformation_net.forward_views(point, times)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters() + list(self.timenet.parameters())

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()
