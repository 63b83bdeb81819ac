"""HexPlane feature field of the 4D-Gaussians deformation network on the MI355X library -- the counterpart of the reference's
gaussian_splatting/utils/hexplane.py with the same public names (HexPlaneField, init_grid_param, interpolate_ms_features,
normalize_aabb), the same parameter names / logical shapes (so a reference state dict loads unchanged) and the same values,
but ONE fused HIP launch per direction instead of 24 F.grid_sample calls (include/deformation_field.h, csrc/gs_hexplane.h).

Layout: every plane keeps the reference's logical shape [1, C, res[c1], res[c0]] (hexplane.py:66-68) but is stored
channels_last -- physically [H][W][C] -- so that the C features of a texel are one contiguous run.  Everything that works on the
logical shape (load_state_dict, the TV regularisers of scene/gaussian_model.py:980-1031, optimizers) is unaffected.

There is no CPU path: the product fails loudly when the tensors are not on a HIP device or the library is missing."""
import ctypes
import itertools
import os
from typing import Iterable, Optional, Sequence

import torch
import torch.nn as nn

from diff_gaussian_rasterization import _C

MAX_LEVELS = 8
BINNED_MIN_POINTS = 49152        # below this the direct-atomic backward is as fast (both are launch-bound) and needs no workspace


class _Level(ctypes.Structure):
    _fields_ = [("planes", ctypes.c_void_p * 6), ("grad_planes", ctypes.c_void_p * 6), ("res", ctypes.c_int32 * 4)]


class _Field(ctypes.Structure):
    _fields_ = [("num_levels", ctypes.c_int32), ("feat_dim", ctypes.c_int32), ("channels_last", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("aabb", ctypes.c_void_p), ("levels", _Level * MAX_LEVELS)]


_lib_cache = None


def _lib():
    global _lib_cache
    if _lib_cache is None:
        lib = _C.load_library()
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        lib.gsr_hexplane_forward.restype = ctypes.c_int
        lib.gsr_hexplane_forward.argtypes = [ctypes.POINTER(_Field), i64, vp, i64, vp, i64, vp, vp]
        lib.gsr_hexplane_backward.restype = ctypes.c_int
        lib.gsr_hexplane_backward.argtypes = [ctypes.POINTER(_Field), i64, vp, i64, vp, i64, vp, vp, vp, vp]
        lib.gsr_hexplane_backward_workspace_size.restype = ctypes.c_size_t
        lib.gsr_hexplane_backward_workspace_size.argtypes = [ctypes.POINTER(_Field), i64]
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.gsr_hexplane_forward_views.restype = ctypes.c_int
        lib.gsr_hexplane_forward_views.argtypes = [ctypes.POINTER(_Field), i64, vp, i64, ctypes.c_int, f32p, vp, vp]
        lib.gsr_hexplane_backward_views_workspace_size.restype = ctypes.c_size_t
        lib.gsr_hexplane_backward_views_workspace_size.argtypes = [ctypes.POINTER(_Field), i64, ctypes.c_int]
        lib.gsr_hexplane_backward_views.restype = ctypes.c_int
        lib.gsr_hexplane_backward_views.argtypes = [ctypes.POINTER(_Field), i64, vp, i64, ctypes.c_int, f32p, vp, vp, vp, vp, vp]
        _lib_cache = lib
    return _lib_cache


def _plane_layout(p: torch.Tensor) -> int:
    """1 = channels_last ([H][W][C] memory), 0 = contiguous ([C][H][W]); anything else is rejected."""
    _, C, H, W = p.shape
    s = p.stride()
    if (C == 1 or s[1] == 1) and (W == 1 or s[3] == C) and (H == 1 or s[2] == W * C):
        return 1
    if (W == 1 or s[3] == 1) and (H == 1 or s[2] == W) and (C == 1 or s[1] == H * W):
        return 0
    raise ValueError(f"HexPlane plane of shape {tuple(p.shape)} has strides {s}: neither channels_last nor contiguous")


_descriptor_cache = {}


def _describe(levels, aabb, grads=None):
    """levels: list of lists of six [1, C, H, W] tensors -> the C descriptor.  The validated geometry of a plane set is cached by
    the planes' (address, shape, strides): a field is described once, later calls only refresh the aabb and gradient pointers."""
    key = tuple((p.data_ptr(), p.shape, p.stride()) for planes in levels for p in planes)
    cached = _descriptor_cache.get(key)
    if cached is None:
        if len(_descriptor_cache) > 64:
            _descriptor_cache.clear()
        cached = _descriptor_cache[key] = _describe_uncached(levels)
    f = _Field.from_buffer_copy(cached)
    f.aabb = aabb.data_ptr() if aabb is not None else None
    if grads is not None:
        for l, row in enumerate(grads):
            for p, g in enumerate(row):
                f.levels[l].grad_planes[p] = g.data_ptr() if g is not None else None
    return f


def _describe_uncached(levels):
    aabb, grads = None, None
    if not 1 <= len(levels) <= MAX_LEVELS:
        raise ValueError(f"HexPlane field with {len(levels)} levels (1..{MAX_LEVELS} supported)")
    f = _Field()
    f.num_levels = len(levels)
    C = levels[0][0].shape[1]
    f.feat_dim = C
    layout = _plane_layout(levels[0][0])
    f.channels_last = layout
    f.aabb = aabb.data_ptr() if aabb is not None else None
    combos = list(itertools.combinations(range(4), 2))
    for l, planes in enumerate(levels):
        if len(planes) != 6:
            raise ValueError("a HexPlane level has six planes (grid_dimensions=2 over 4 input coordinates)")
        res = [0, 0, 0, 0]
        for (c0, c1), p in zip(combos, planes):
            _C._require_device(p, "HexPlane plane")
            if p.dtype != torch.float32 or p.dim() != 4 or p.shape[0] != 1 or p.shape[1] != C:
                raise ValueError(f"HexPlane plane must be fp32 [1, {C}, H, W], got {p.dtype} {tuple(p.shape)}")
            if _plane_layout(p) != layout:
                raise ValueError("all HexPlane planes must share one memory layout")
            for c, size in ((c0, p.shape[3]), (c1, p.shape[2])):      # first coordinate indexes the width (hexplane.py:66-68)
                if res[c] not in (0, size):
                    raise ValueError(f"level {l}: inconsistent resolution along coordinate {c}: {res[c]} vs {size}")
                res[c] = size
        for p in range(6):
            f.levels[l].planes[p] = planes[p].data_ptr()
            f.levels[l].grad_planes[p] = grads[l][p].data_ptr() if grads is not None and grads[l][p] is not None else None
        for k in range(4):
            f.levels[l].res[k] = res[k]
    return bytes(f)


class _HexPlaneFeatures(torch.autograd.Function):
    """features [n, L*C] of points xyz [n, >=3] (strided rows allowed) at times [n, >=1]."""

    @staticmethod
    def forward(ctx, xyz, time, aabb, n_levels, *planes):
        glue = _C._glue
        if glue is not None and hasattr(glue, "hexplane_forward"):        # native host glue (csrc/torch_glue.cpp)
            _C._require_device(xyz, "pts")
            if xyz.dim() == 2 and xyz.stride(1) != 1:
                xyz = xyz.contiguous()
            det = [p.detach() for p in planes]
            try:
                out = glue.hexplane_forward(det, n_levels, xyz.detach(), time.detach(), None if aabb is None else aabb.detach(), _C._stream(xyz.device))
            except (ValueError, RuntimeError) as e:
                if isinstance(e, ValueError) or "HIP device" in str(e):
                    raise
                raise ValueError(str(e)) from e
            ctx.save_for_backward(xyz, time, aabb if aabb is not None else torch.empty(0), *planes)
            ctx.n_levels, ctx.has_aabb, ctx.use_glue = n_levels, aabb is not None, True
            return out
        ctx.use_glue = False
        _C._require_device(xyz, "pts")
        _C._require_device(time, "timestamps")
        if xyz.dtype != torch.float32 or time.dtype != torch.float32:
            raise ValueError("HexPlane inputs must be fp32")
        if xyz.dim() != 2 or xyz.shape[1] < 3 or time.dim() != 2 or time.shape[0] != xyz.shape[0]:
            raise ValueError(f"HexPlane expects pts [n, 3] and timestamps [n, 1], got {tuple(xyz.shape)} and {tuple(time.shape)}")
        if xyz.stride(1) != 1:
            xyz = xyz.contiguous()
        levels = [list(planes[6 * l:6 * l + 6]) for l in range(n_levels)]
        levels = [[p.detach() for p in lv] for lv in levels]
        n, C = xyz.shape[0], levels[0][0].shape[1]
        out = torch.empty((n, n_levels * C), dtype=torch.float32, device=xyz.device)
        field = _describe(levels, aabb)
        lib = _lib()
        with torch.cuda.device(xyz.device):
            rc = lib.gsr_hexplane_forward(ctypes.byref(field), n, xyz.data_ptr(), xyz.stride(0), time.data_ptr(), time.stride(0),
                                          out.data_ptr(), _C._stream(xyz.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_hexplane_forward")
        ctx.save_for_backward(xyz, time, aabb if aabb is not None else torch.empty(0), *planes)
        ctx.n_levels, ctx.has_aabb = n_levels, aabb is not None
        return out

    @staticmethod
    def backward(ctx, g):
        xyz, time, aabb, *planes = ctx.saved_tensors
        n_levels = ctx.n_levels
        aabb = aabb if ctx.has_aabb else None
        if ctx.use_glue:
            need_plane = list(ctx.needs_input_grad[4:])
            mode = os.environ.get("GSR_HEX_BINNED", "auto")
            sorted_bwd = any(need_plane) and mode != "0" and (mode == "1" or xyz.shape[0] >= BINNED_MIN_POINTS)
            res = _C._glue.hexplane_backward([p.detach() for p in planes], n_levels, xyz, time, aabb, g, need_plane,
                                             bool(ctx.needs_input_grad[0]), sorted_bwd, _C._stream(g.device))
            gxyz = res[0]
            if gxyz is not None and xyz.shape[1] > 3:
                full = torch.zeros_like(xyz)
                full[:, :3] = gxyz
                gxyz = full
            return (gxyz, None, None, None, *res[1:])
        levels = [[p.detach() for p in planes[6 * l:6 * l + 6]] for l in range(n_levels)]
        need_plane = list(ctx.needs_input_grad[4:])
        layout = _plane_layout(levels[0][0])
        # one zeroed buffer for every plane gradient (one memset instead of 24); each gradient is a view with the plane's strides
        sizes = [p.numel() if need else 0 for p, need in zip(planes, need_plane)]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=g.device)
        grads, views, o = [], [], 0
        for p, need, sz in zip(planes, need_plane, sizes):
            if not need:
                views.append(None)
                continue
            views.append(torch.as_strided(flat, p.shape, p.stride(), o))      # the plane's own strides (either layout), offset o
            o += sz
        grads = [views[6 * l:6 * l + 6] for l in range(n_levels)]
        g = g.contiguous()
        gxyz = torch.empty((xyz.shape[0], 3), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        field = _describe(levels, aabb, grads)
        lib = _lib()
        # large batches on channels-last planes: the binned algorithm (counting sort per plane family, LDS accumulation per plane
        # region, one flush) -- needs a workspace; small ones: one atomic per (point, corner)
        ws = None
        mode = os.environ.get("GSR_HEX_BINNED", "auto")
        if layout == 1 and any(need_plane) and mode != "0" and (mode == "1" or xyz.shape[0] >= BINNED_MIN_POINTS):
            ws = torch.empty(lib.gsr_hexplane_backward_workspace_size(ctypes.byref(field), xyz.shape[0]), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.gsr_hexplane_backward(ctypes.byref(field), xyz.shape[0], xyz.data_ptr(), xyz.stride(0), time.data_ptr(), time.stride(0),
                                           g.data_ptr(), gxyz.data_ptr() if gxyz is not None else None,
                                           ws.data_ptr() if ws is not None else None, _C._stream(g.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_hexplane_backward")
        if gxyz is not None and xyz.shape[1] > 3:                  # rows were a slice of a wider tensor
            full = torch.zeros_like(xyz)
            full[:, :3] = gxyz
            gxyz = full
        return (gxyz, None, None, None, *views)


MAX_VIEWS = 12          # GSR_HEXPLANE_MAX_VIEWS


class _HexPlaneFeaturesViews(torch.autograd.Function):
    """features [V, n, L*C] of the SAME points xyz [n, >=3] at V times (Python floats): the field of every keyframe of one mapping
    iteration (gaussian_renderer/__init__.py:112,149-157) in one launch per direction -- the spatial planes gathered once per point, one
    counting sort and one spatial scatter for all views (include/deformation_field.h gsr_hexplane_*_views). View v is bit-identical to
    _HexPlaneFeatures at time times[v]; the gradients are those of V single calls up to summation order."""

    @staticmethod
    def forward(ctx, xyz, times, aabb, n_levels, *planes):
        _C._require_device(xyz, "pts")
        if xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.shape[1] < 3:
            raise ValueError(f"HexPlane expects fp32 pts [n, 3], got {xyz.dtype} {tuple(xyz.shape)}")
        if xyz.stride(1) != 1:
            xyz = xyz.contiguous()
        V = len(times)
        if not 1 <= V <= MAX_VIEWS:
            raise ValueError(f"1..{MAX_VIEWS} times expected, got {V}")
        levels = [[p.detach() for p in planes[6 * l:6 * l + 6]] for l in range(n_levels)]
        n, C = xyz.shape[0], levels[0][0].shape[1]
        out = torch.empty((V, n, n_levels * C), dtype=torch.float32, device=xyz.device)
        field = _describe(levels, aabb)
        tv = (ctypes.c_float * V)(*[float(t) for t in times])
        lib = _lib()
        with torch.cuda.device(xyz.device):
            rc = lib.gsr_hexplane_forward_views(ctypes.byref(field), n, xyz.data_ptr(), xyz.stride(0), V, tv, out.data_ptr(), _C._stream(xyz.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_hexplane_forward_views")
        ctx.save_for_backward(xyz, aabb if aabb is not None else torch.empty(0), *planes)
        ctx.n_levels, ctx.has_aabb, ctx.times = n_levels, aabb is not None, tv
        return out

    @staticmethod
    def backward(ctx, g):
        xyz, aabb, *planes = ctx.saved_tensors
        n_levels, V, n = ctx.n_levels, len(ctx.times), xyz.shape[0]
        aabb = aabb if ctx.has_aabb else None
        levels = [[p.detach() for p in planes[6 * l:6 * l + 6]] for l in range(n_levels)]
        need_plane = list(ctx.needs_input_grad[4:])
        sizes = [p.numel() if need else 0 for p, need in zip(planes, need_plane)]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=g.device)          # one fill for every plane gradient
        views, o = [], 0
        for p, need, sz in zip(planes, need_plane, sizes):
            views.append(torch.as_strided(flat, p.shape, p.stride(), o) if need else None)
            o += sz
        grads = [views[6 * l:6 * l + 6] for l in range(n_levels)]
        g = g.contiguous()
        gxyz = torch.empty((n, 3), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        field = _describe(levels, aabb, grads)
        lib = _lib()
        size = lib.gsr_hexplane_backward_views_workspace_size(ctypes.byref(field), n, V) if n else 0
        if n and size == 0:
            raise RuntimeError("hexplane_features_views: this plane geometry is not covered by the batched backward (channels-last planes, "
                               "resolutions <= 1024); evaluate the views one by one with hexplane_features")
        if n:
            ws = torch.empty(size, dtype=torch.uint8, device=g.device)
            with torch.cuda.device(g.device):
                rc = lib.gsr_hexplane_backward_views(ctypes.byref(field), n, xyz.data_ptr(), xyz.stride(0), V, ctx.times, g.data_ptr(), None,
                                                     gxyz.data_ptr() if gxyz is not None else None, ws.data_ptr(), _C._stream(g.device))
            if rc < 0:
                _C._err(lib, rc, "gsr_hexplane_backward_views")
        if gxyz is not None and xyz.shape[1] > 3:
            full = torch.zeros_like(xyz)
            full[:, :3] = gxyz
            gxyz = full
        return (gxyz, None, None, None, *views)


def views_supported(ms_grids, n_views) -> bool:
    """Whether hexplane_features_views covers this field: channels-last planes of at most 1024 texels a side, 1..MAX_VIEWS views."""
    flat = ms_grids.flat if isinstance(ms_grids, _PlaneList) else [p for lv in ms_grids for p in lv]
    try:
        return (1 <= n_views <= MAX_VIEWS and all(p.is_cuda and p.dtype == torch.float32 and _plane_layout(p) == 1 and max(p.shape[2:]) <= 1024 for p in flat)
                and flat[0].shape[1] in (8, 16, 32, 64) and not (flat[0].shape[1] == 64 and len(flat) > 24))
    except ValueError:
        return False


def hexplane_features_views(pts, times, aabb, ms_grids) -> torch.Tensor:
    """[V, n, L*C]: hexplane_features(pts, full((n, 1), times[v]), aabb, ms_grids) for every v, batched (see _HexPlaneFeaturesViews)."""
    if isinstance(ms_grids, _PlaneList):
        return _HexPlaneFeaturesViews.apply(pts, tuple(times), aabb, ms_grids.n_levels, *ms_grids.flat)
    levels = [list(g) for g in ms_grids]
    return _HexPlaneFeaturesViews.apply(pts, tuple(times), aabb, len(levels), *[p for lv in levels for p in lv])


def hexplane_features(pts, timestamps, aabb, ms_grids) -> torch.Tensor:
    """Fused normalize_aabb + interpolate_ms_features(concat_features=True).  aabb None: pts are already normalised."""
    if isinstance(ms_grids, _PlaneList):                              # prepared by HexPlaneField: no nn.ParameterList walk per call
        return _HexPlaneFeatures.apply(pts, timestamps, aabb, ms_grids.n_levels, *ms_grids.flat)
    levels = [list(g) for g in ms_grids]
    flat = [p for lv in levels for p in lv]
    return _HexPlaneFeatures.apply(pts, timestamps, aabb, len(levels), *flat)


class _PlaneList:
    """The planes of a HexPlaneField's nn.ModuleList of nn.ParameterList as a flat Python list."""

    def __init__(self, grids):
        # straight from the registries: iterating nn.ModuleList / nn.ParameterList goes through __getitem__ (~4 us per entry)
        self.flat = [p for lv in grids._modules.values() for p in lv._parameters.values()]
        self.n_levels = len(grids._modules)

    def __iter__(self):                                               # still usable as ms_grids: a sequence of levels
        return (self.flat[6 * l:6 * l + 6] for l in range(self.n_levels))


# ---- the reference module's public names ------------------------------------------------------------------------------------

def normalize_aabb(pts, aabb):
    """hexplane.py:19-22 (the fused field applies this inside the kernel; provided for callers that use it on its own)."""
    aabb = aabb.to(device=pts.device)
    return torch.clamp((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, -1.0, 1.0)


def init_grid_param(grid_nd: int, in_dim: int, out_dim: int, reso: Sequence[int], a: float = 0.1, b: float = 0.5):
    """hexplane.py:52-76: one parameter per coordinate pair, [1, out_dim, reso[c1], reso[c0]]; planes that include the time axis
    start at 1, the others uniform in [a, b].  Stored channels_last."""
    if in_dim != len(reso):
        raise AssertionError("Resolution must have same number of elements as input-dimension")
    if grid_nd != 2 or in_dim != 4:
        raise NotImplementedError("the MI355X HexPlane field implements the shipped geometry: 2-D planes over (x, y, z, t)")
    coefs = nn.ParameterList()
    for c0, c1 in itertools.combinations(range(in_dim), grid_nd):
        p = torch.empty([1, out_dim, reso[c1], reso[c0]]).contiguous(memory_format=torch.channels_last)
        if 3 in (c0, c1):
            nn.init.ones_(p)
        else:
            nn.init.uniform_(p, a=a, b=b)
        coefs.append(nn.Parameter(p))
    return coefs


def interpolate_ms_features(pts: torch.Tensor, ms_grids: Iterable, grid_dimensions: int = 2, concat_features: bool = True,
                            num_levels: Optional[int] = None) -> torch.Tensor:
    """hexplane.py:81-112 for already-normalised pts [n, 4] = (x, y, z, t)."""
    if grid_dimensions != 2 or pts.shape[-1] != 4:
        raise NotImplementedError("the MI355X HexPlane field implements 2-D planes over 4 input coordinates")
    grids = list(ms_grids)[:num_levels] if num_levels is not None else list(ms_grids)
    feats = hexplane_features(pts[:, :3], pts[:, 3:4], None, grids)
    if concat_features:
        return feats
    C = grids[0][0].shape[1]
    return feats.view(feats.shape[0], len(grids), C).sum(dim=1)       # the sum-over-levels variant (:107-108)


class HexPlaneField(nn.Module):
    """hexplane.py:115-188: same constructor, attributes (aabb, grids, feat_dim, grid_config, multiscale_res_multipliers,
    concat_features), state-dict keys and forward value."""

    def __init__(self, bounds, planeconfig, multires) -> None:
        super().__init__()
        self.aabb = nn.Parameter(torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]], dtype=torch.float32),
                                 requires_grad=False)                   # [0] is the max corner (:127-129): the field is mirrored
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        for res in self.multiscale_res_multipliers:
            config = self.grid_config[0].copy()
            config["resolution"] = [r * res for r in config["resolution"][:3]] + config["resolution"][3:]   # spatial axes only (:139-142)
            gp = init_grid_param(grid_nd=config["grid_dimensions"], in_dim=config["input_coordinate_dim"],
                                 out_dim=config["output_coordinate_dim"], reso=config["resolution"])
            self.feat_dim = self.feat_dim + gp[-1].shape[1] if self.concat_features else gp[-1].shape[1]
            self.grids.append(gp)

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        self.aabb = nn.Parameter(torch.tensor([xyz_max, xyz_min], dtype=torch.float32).to(self.aabb.device), requires_grad=False)

    def get_density(self, pts: torch.Tensor, timestamps: Optional[torch.Tensor] = None):
        pts = pts.reshape(-1, pts.shape[-1])
        if pts.shape[0] == 0:
            return torch.zeros((0, 1), device=pts.device)              # :174-175
        return hexplane_features(pts, timestamps.reshape(-1, timestamps.shape[-1]), self.aabb, _PlaneList(self.grids))

    def forward(self, pts: torch.Tensor, timestamps: Optional[torch.Tensor] = None):
        return self.get_density(pts, timestamps)

    def forward_views(self, pts: torch.Tensor, times):
        """[V, n, feat_dim]: forward(pts, time = times[v]) for the V keyframes of one mapping iteration in one launch; None when the
        batched kernels do not cover this field (the caller then evaluates view by view)."""
        planes = _PlaneList(self.grids)
        pts = pts.reshape(-1, pts.shape[-1])
        if not views_supported(planes, len(times)) or pts.shape[0] == 0 or not pts.is_cuda:
            return None
        return hexplane_features_views(pts, times, self.aabb, planes)
