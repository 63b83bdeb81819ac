"""SC-GS control-node warp on the MI355X library (include/control_nodes.h) -- the per-Gaussian half of the reference's
utils/time_utils.py ControlNodeWarp: `knn_points` (the pytorch3d.ops routine the reference imports, which has no ROCm build),
`cal_nn_weight` (:981-1011) and `node_blend`, the body of ControlNodeWarp.forward (:1192-1258) after the node MLP.

The node MLP stays in torch (O(nodes) library GEMMs).  Everything else -- the K nearest nodes, exp / sigmoid of the raw node
radius / weight (:893-898), quaternion_to_matrix of the local rotations (:115-133,1207-1208), the RBF weights, the blend, and all
of their chain rules -- is one HIP launch forward and three backward.  There is no CPU path."""
import ctypes
from collections import namedtuple

import torch

from diff_gaussian_rasterization import _C

MAX_K, MAX_DIM, BLEND_MAX_K = 32, 32, 8
_KNN = namedtuple("KNN", "dists idx knn")


class _Blend(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("m", ctypes.c_int32), ("K", ctypes.c_int32), ("local_frame", ctypes.c_int32),
                ("rot_as_residual", ctypes.c_int32), ("node_stride", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("motion_mask", ctypes.c_void_p), ("nodes", ctypes.c_void_p), ("node_radius", ctypes.c_void_p),
                ("node_weight", ctypes.c_void_p), ("node_trans", ctypes.c_void_p), ("node_rot", ctypes.c_void_p),
                ("node_scale", ctypes.c_void_p), ("node_frame", ctypes.c_void_p), ("node_local_rotation", ctypes.c_void_p),
                ("attr_stride", ctypes.c_int32), ("grad_stride", ctypes.c_int32)]


_lib_cache = None


def _lib():
    global _lib_cache
    if _lib_cache is None:
        lib = _C.load_library()
        i64, vp, i = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        lib.gsr_knn_points.restype = i
        lib.gsr_knn_points.argtypes = [i64, i64, i, i, vp, vp, vp, vp, vp]
        lib.gsr_knn_points_batch.restype = i
        lib.gsr_knn_points_batch.argtypes = [i64, i64, i64, i, i, vp, vp, vp, vp, vp]
        lib.gsr_node_blend_forward.restype = i
        lib.gsr_node_blend_forward.argtypes = [ctypes.POINTER(_Blend), vp, vp, vp, vp, vp, vp, vp]
        lib.gsr_node_blend_workspace_size.restype = ctypes.c_size_t
        lib.gsr_node_blend_workspace_size.argtypes = [i64, ctypes.c_int32]
        lib.gsr_node_blend_backward.restype = i
        lib.gsr_node_blend_backward.argtypes = [ctypes.POINTER(_Blend)] + [vp] * 15
        lib.gsr_node_blend_forward_batch.restype = i
        lib.gsr_node_blend_forward_batch.argtypes = [ctypes.POINTER(_Blend), i, vp, vp, vp, vp, vp, vp, vp]
        lib.gsr_node_blend_workspace_size_batch.restype = ctypes.c_size_t
        lib.gsr_node_blend_workspace_size_batch.argtypes = [i64, ctypes.c_int32, i]
        lib.gsr_node_blend_backward_batch.restype = i
        lib.gsr_node_blend_backward_batch.argtypes = [ctypes.POINTER(_Blend), i] + [vp] * 15
        lib.gsr_multi_add.restype = i
        lib.gsr_multi_add.argtypes = [i, ctypes.POINTER(_MultiAddItem), vp]
        lib.gsr_index_csr_workspace_size.restype = ctypes.c_size_t
        lib.gsr_index_csr_workspace_size.argtypes = [i, i, i]
        lib.gsr_index_csr.restype = i
        lib.gsr_index_csr.argtypes = [i, i, i, vp, vp, vp]
        lib.gsr_segment_sum.restype = i
        lib.gsr_segment_sum.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp]
        lib.gsr_node_embedding.restype = i
        lib.gsr_node_embedding.argtypes = [i, i, i, i, vp, i, vp, vp, vp, vp]
        lib.gsr_node_embedding_workspace_size.restype = ctypes.c_size_t
        lib.gsr_node_embedding_workspace_size.argtypes = [i, i, i, i]
        lib.gsr_relu_backward_bias_workspace_size.restype = ctypes.c_size_t
        lib.gsr_relu_backward_bias_workspace_size.argtypes = [i, i]
        lib.gsr_relu_backward_bias.restype = i
        lib.gsr_relu_backward_bias.argtypes = [i, i, vp, vp, vp, vp, vp, vp]
        _lib_cache = lib
    return _lib_cache


def _f32(t, name):
    _C._require_device(t, name)
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be fp32, got {t.dtype}")
    return t.detach().contiguous()


def knn_points(p1, p2, lengths1=None, lengths2=None, K: int = 1, version: int = -1, return_nn: bool = False, return_sorted: bool = True):
    """pytorch3d.ops.knn_points for equal-length batches: p1 [B, N, D], p2 [B, M, D] -> (dists [B, N, K] squared, idx [B, N, K]
    int64, knn [B, N, K, D] or None).  No gradient flows through dists (the reference only uses detached inputs here)."""
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("knn_points: ragged batches (lengths1 / lengths2) are not used by the reference and not implemented")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != p2.shape[2]:
        raise ValueError(f"knn_points expects p1 [B, N, D] and p2 [B, M, D], got {tuple(p1.shape)} and {tuple(p2.shape)}")
    B, N, D = p1.shape
    if not (1 <= K <= MAX_K) or not (1 <= D <= MAX_DIM):
        raise ValueError(f"knn_points: K = {K}, D = {D} outside 1..32")
    a, b = _f32(p1, "p1"), _f32(p2, "p2")
    dists = torch.empty((B, N, K), dtype=torch.float32, device=a.device)
    idx = torch.empty((B, N, K), dtype=torch.int64, device=a.device)
    lib = _lib()
    with torch.cuda.device(a.device):
        rc = lib.gsr_knn_points_batch(B, N, b.shape[1], D, K, a.data_ptr(), b.data_ptr(), dists.data_ptr(), idx.data_ptr(), _C._stream(a.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_knn_points_batch")
    knn = None
    if return_nn:
        knn = torch.gather(p2[:, None].expand(-1, N, -1, -1), 2, idx[..., None].expand(-1, -1, -1, D))
    return _KNN(dists, idx, knn)


class _MultiAddItem(ctypes.Structure):      # gsr_multi_add_item
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p * 4), ("count", ctypes.c_int32)]


class _FanOut(torch.autograd.Function):
    """outputs[k] = stacked[plan[k][0]][plan[k][1]] (aliases: no copy); backward: every row of every stacked tensor's gradient = the sum of the
    gradients of its readers, all rows in ONE launch (gsr_multi_add)."""

    @staticmethod
    def forward(ctx, plan, *stacked):
        ctx.plan, ctx.shapes = plan, [tuple(t.shape) for t in stacked]
        ctx.set_materialize_grads(False)
        return tuple(stacked[a].detach()[i] for a, i in plan)

    @staticmethod
    def backward(ctx, *grads):
        plan, shapes = ctx.plan, ctx.shapes
        dev = next(g for g in grads if g is not None).device if any(g is not None for g in grads) else None
        if dev is None:
            return (None,) + (None,) * len(shapes)
        outs = [torch.empty(shape, dtype=torch.float32, device=dev) for shape in shapes]
        readers, keep = {}, []
        for (a, i), g in zip(plan, grads):
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.to(torch.float32).contiguous()
            keep.append(g)
            readers.setdefault((a, i), []).append(g)
        rows = [(a, i) for a, shape in enumerate(shapes) for i in range(shape[0])]
        lib = _lib()
        with torch.cuda.device(dev):
            for lo in range(0, len(rows), 64):
                part = rows[lo:lo + 64]
                items = (_MultiAddItem * len(part))()
                for it, (a, i) in zip(items, part):
                    srcs = readers.get((a, i), [])
                    if len(srcs) > 4:                     # (more readers than a launch item carries: fold the rest first)
                        extra = srcs[3]
                        for g in srcs[4:]:
                            extra = extra + g
                        keep.append(extra)
                        srcs = srcs[:3] + [extra]
                    it.dst, it.count = outs[a][i].data_ptr(), int(outs[a][i].numel())
                    for k in range(4):
                        it.src[k] = srcs[k].data_ptr() if k < len(srcs) else None
                rc = lib.gsr_multi_add(len(part), items, _C._stream(dev))
                if rc < 0:
                    _C._err(lib, rc, "gsr_multi_add")
        return (None,) + tuple(outs)


def fan_out(stacked, plan):
    """Rows of stacked tensors for SEVERAL readers each: returns [stacked[a][i] for (a, i) in plan] (the same row may appear many times) as one
    autograd node whose backward pass adds the readers' gradients of all rows in one launch and hands back whole stacked gradients -- instead
    of an unbind per tensor, a pairwise addition per extra reader and a re-stacking (32 + 3 launches of tiny kernels per dynamic mapping
    iteration, slam/dynamic_graph.py). stacked: fp32 device tensors [S, ...]; rows nobody reads get zero gradients."""
    for t in stacked:
        _C._require_device(t, "stacked")
    return list(_FanOut.apply(tuple((int(a), int(i)) for a, i in plan), *[t.contiguous() for t in stacked]))


class IndexSets:
    """S index sets idx [S, E] (int64, values in [0, Nv)) together with their reverse lists (gsr_index_csr): what gather_rows needs to
    run its backward pass as an ORDERED segment sum. Build it once per index array, use it for every gather through that array."""

    def __init__(self, idx, n_targets):
        _C._require_device(idx, "idx")
        if idx.dtype != torch.int64 or idx.dim() != 2:
            raise ValueError("IndexSets: idx must be an int64 [S, E] device tensor")
        self.idx = idx.contiguous()
        self.S, self.E, self.Nv = int(idx.shape[0]), int(idx.shape[1]), int(n_targets)
        lib = _lib()
        self.csr = torch.empty((int(lib.gsr_index_csr_workspace_size(self.S, self.E, self.Nv)),), dtype=torch.uint8, device=idx.device)
        with torch.cuda.device(idx.device):
            rc = lib.gsr_index_csr(self.S, self.E, self.Nv, self.idx.data_ptr(), self.csr.data_ptr(), _C._stream(idx.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_index_csr")


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, sets, set_of_b):
        # table [B, Nv, C]; out[b, e, :] = table[b, idx[set_of_b[b], e], :]
        B, Nv, Cn = table.shape
        idx = sets.idx if set_of_b is None else sets.idx.index_select(0, getattr(set_of_b, "_gsr_long", None) if hasattr(set_of_b, "_gsr_long") else set_of_b.long())
        idx = idx if idx.shape[0] == B else idx.expand(B, -1)
        ctx.sets, ctx.set_of_b, ctx.shape = sets, set_of_b, (B, Nv, Cn)
        return torch.gather(table, 1, idx[:, :, None].expand(-1, -1, Cn))

    @staticmethod
    def backward(ctx, g):
        sets, (B, Nv, Cn) = ctx.sets, ctx.shape
        g = g.to(torch.float32).contiguous()
        out = torch.empty((B, Nv, Cn), dtype=torch.float32, device=g.device)
        lib = _lib()
        sob = None if ctx.set_of_b is None else ctx.set_of_b.to(torch.int32).contiguous()
        with torch.cuda.device(g.device):
            rc = lib.gsr_segment_sum(B, sets.S, sets.E, Cn, Nv, g.data_ptr(), sets.csr.data_ptr(), None if sob is None else sob.data_ptr(), out.data_ptr(),
                                     _C._stream(g.device))
        if rc < 0:
            _C._err(lib, rc, "gsr_segment_sum")
        return out, None, None


def gather_rows(table, sets: IndexSets, set_of_b=None):
    """out[b, e, :] = table[b, idx[s, e], :] with s = set_of_b[b] (int tensor [B]; None: the one set, or one set per batch element when
    S == B) -- torch.gather's values, but a BACKWARD pass that adds the incoming rows of a target in a fixed order (gsr_segment_sum) instead of
    torch's scatter_add with float atomics: bit-reproducible gradients. table [B, Nv, C] fp32 on the device."""
    if table.dim() != 3 or table.shape[1] != sets.Nv:
        raise ValueError(f"gather_rows: table {tuple(table.shape)} does not match the index sets ({sets.Nv} targets)")
    if set_of_b is None and sets.S not in (1, table.shape[0]):
        raise ValueError("gather_rows: give set_of_b when the number of index sets is neither 1 nor the batch size")
    if set_of_b is None and sets.S == table.shape[0] and sets.S > 1:
        set_of_b = torch.arange(sets.S, device=table.device, dtype=torch.int32)
    return _GatherRows.apply(table, sets, set_of_b)


def node_embedding(nodes, times, n_freq_x, n_freq_t):
    """[n * M, 3 (1 + 2 Fx) + 1 + 2 Ft]: the node network's input for every (time sample, node) pair in one launch (gsr_node_embedding,
    include/control_nodes.h). nodes [M, 3] and times [n] fp32 on the device; no gradient (node positions are detached, times are data)."""
    _C._require_device(nodes, "nodes")
    nodes, times = _f32(nodes, "nodes"), _f32(times.reshape(-1), "times")
    if nodes.dim() != 2 or nodes.shape[1] < 3:
        raise ValueError(f"node_embedding expects nodes [M, >=3], got {tuple(nodes.shape)}")
    n, M = int(times.shape[0]), int(nodes.shape[0])
    out = torch.empty((n * M, 3 * (1 + 2 * n_freq_x) + 1 + 2 * n_freq_t), dtype=torch.float32, device=nodes.device)
    lib = _lib()
    ws = torch.empty((int(lib.gsr_node_embedding_workspace_size(n, M, int(n_freq_x), int(n_freq_t))),), dtype=torch.uint8, device=nodes.device)
    with torch.cuda.device(nodes.device):
        rc = lib.gsr_node_embedding(n, M, int(n_freq_x), int(n_freq_t), nodes.data_ptr(), int(nodes.shape[1]), times.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                    _C._stream(nodes.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_node_embedding")
    return out


RELU_BIAS_COLS = (64, 128, 256, 512, 1024)


def relu_backward_bias(dY, Y):
    """(G, dbias) of a layer y = relu(x W^T + b): G = dY * (Y > 0) and dbias = G.sum(0), in one pass over dY and Y (gsr_relu_backward_bias,
    include/control_nodes.h) with the column sums formed in a fixed order. dY, Y [rows, cols] fp32 on the device, cols in RELU_BIAS_COLS."""
    _C._require_device(dY, "dY")
    if dY.dim() != 2 or dY.shape != Y.shape or dY.dtype != torch.float32 or Y.dtype != torch.float32 or int(dY.shape[1]) not in RELU_BIAS_COLS:
        raise ValueError(f"relu_backward_bias expects two fp32 [rows, cols] tensors with cols in {RELU_BIAS_COLS}, got {tuple(dY.shape)} and {tuple(Y.shape)}")
    dY, Y = dY.contiguous(), Y.contiguous()
    rows, cols = int(dY.shape[0]), int(dY.shape[1])
    G = torch.empty_like(dY)
    db = torch.empty((cols,), dtype=torch.float32, device=dY.device)
    lib = _lib()
    ws = torch.empty((max(16, int(lib.gsr_relu_backward_bias_workspace_size(rows, cols))),), dtype=torch.uint8, device=dY.device)
    with torch.cuda.device(dY.device):
        rc = lib.gsr_relu_backward_bias(rows, cols, dY.data_ptr(), Y.data_ptr(), G.data_ptr(), db.data_ptr(), ws.data_ptr(), _C._stream(dY.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_relu_backward_bias")
    return G, db


def quaternion_to_matrix(q):
    """utils/time_utils.py:115-133: real part first; the 2 / |q|^2 factor normalises."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


RADIUS_IS_LOG, WEIGHT_IS_LOGIT = 1, 2


class _NodeBlend(torch.autograd.Function):
    """(nn_weight, nn_dist, nn_idx, d_xyz, d_rotation, d_scaling) from per-node tensors; gradients to node_radius, node_weight,
    node_trans, node_rot, node_scale, local_rotation.  x and nodes are constants of the op (detached in the reference)."""

    @staticmethod
    def forward(ctx, x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation, K,
                rot_as_residual, raw):
        if not 1 <= K <= BLEND_MAX_K:
            raise ValueError(f"node blend: K = {K} outside 1..{BLEND_MAX_K}")
        glue = _C._glue
        if glue is not None and hasattr(glue, "node_blend_forward"):      # native host glue (csrc/torch_glue.cpp): ~10x less host time
            _C._require_device(x, "x")
            det = lambda t: None if t is None else t.detach()
            args = (x.detach(), det(motion_mask), nodes.detach(), node_radius.detach(), det(node_weight), det(node_trans), det(node_rot),
                    det(node_scale), det(local_rotation) if node_trans is not None else None, int(K), bool(rot_as_residual),
                    (RADIUS_IS_LOG | WEIGHT_IS_LOGIT) if raw else 0)
            try:
                w, dist, idx, d_xyz, d_rot, d_scale = glue.node_blend_forward(*args, _C._stream(x.device))
            except RuntimeError as e:
                raise ValueError(str(e)) from e
            # (w, dist, idx are OUTPUTS: kept through save_for_backward -- as plain attributes of ctx they form a reference cycle output ->
            # grad_fn -> ctx -> output that only the cyclic collector breaks, and the whole upstream graph, the network's AccumulateGrad
            # nodes included, lingers until then)
            ctx.glue_args = args
            ctx.save_for_backward(w, dist, idx)
            ctx.mark_non_differentiable(dist, idx)
            return w, dist, idx, d_xyz, d_rot, d_scale
        x, nodes, node_radius = _f32(x, "x"), _f32(nodes, "nodes"), _f32(node_radius, "node_radius").reshape(-1)
        if x.dim() != 2 or x.shape[1] != 3 or nodes.dim() != 2 or nodes.shape[1] < 3:
            raise ValueError(f"node blend expects x [N, 3] and nodes [M, >=3], got {tuple(x.shape)} and {tuple(nodes.shape)}")
        n, m = x.shape[0], nodes.shape[0]
        blend = node_trans is not None
        opt = lambda t, name, shape: None if t is None else _checked(_f32(t, name), name, shape)
        motion_mask = opt(motion_mask, "motion_mask", None)
        if motion_mask is not None and motion_mask.numel() != n:
            raise ValueError(f"motion_mask must have one value per Gaussian ({n}), got {tuple(motion_mask.shape)}")
        node_weight = None if node_weight is None else _f32(node_weight, "node_weight").reshape(-1)
        if node_radius.numel() != m or (node_weight is not None and node_weight.numel() != m):
            raise ValueError("node_radius / node_weight must have one value per node")
        node_trans, node_rot = opt(node_trans, "node_trans", (m, 3)), opt(node_rot, "node_rot", (m, 4))
        node_scale = opt(node_scale, "node_scale", (m, 3))
        local_rotation = opt(local_rotation, "local_rotation", (m, 4)) if blend else None
        keep = dict(x=x, motion_mask=motion_mask, nodes=nodes, node_radius=node_radius, node_weight=node_weight, node_trans=node_trans,
                    node_rot=node_rot, node_scale=node_scale, node_local_rotation=local_rotation)
        scalars = dict(n=n, m=m, K=K, local_frame=int(local_rotation is not None), rot_as_residual=int(bool(rot_as_residual)),
                       node_stride=nodes.shape[1], flags=(RADIUS_IS_LOG | WEIGHT_IS_LOGIT) if raw else 0)
        a = _Blend(**scalars)
        for k, t in keep.items():
            setattr(a, k, t.data_ptr() if t is not None else None)
        dev = x.device
        w = torch.empty((n, K), dtype=torch.float32, device=dev)
        dist = torch.empty((n, K), dtype=torch.float32, device=dev)
        idx = torch.empty((n, K), dtype=torch.int64, device=dev)
        outs = [torch.empty((n, c), dtype=torch.float32, device=dev) if blend else None for c in (3, 4, 3)]
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_forward(ctypes.byref(a), w.data_ptr(), dist.data_ptr(), idx.data_ptr(),
                                            *(o.data_ptr() if o is not None else None for o in outs), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_forward")
        ctx.keep, ctx.scalars, ctx.glue_args = keep, scalars, None
        ctx.save_for_backward(w, dist, idx)
        ctx.mark_non_differentiable(dist, idx)
        empty = torch.empty(0, device=dev)
        return (w, dist, idx, *(o if o is not None else empty for o in outs))

    @staticmethod
    def backward(ctx, g_w, _g_dist, _g_idx, g_xyz, g_rot, g_scale):
        if ctx.glue_args is not None:
            w, dist, idx = ctx.saved_tensors
            some = lambda g: g if g is not None and g.numel() else None
            g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local = _C._glue.node_blend_backward(
                *ctx.glue_args, w, dist, idx, some(g_w), some(g_xyz), some(g_rot), some(g_scale), _C._stream(w.device))
            return None, None, None, g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local, None, None, None
        keep, sc = ctx.keep, ctx.scalars
        n, m = sc["n"], sc["m"]
        w, dist, idx = ctx.saved_tensors
        dev = w.device
        blend = keep["node_trans"] is not None
        a = _Blend(**sc)
        for k, t in keep.items():
            setattr(a, k, t.data_ptr() if t is not None else None)
        cot = lambda g: g.contiguous() if g is not None and g.numel() else None
        g_w, g_xyz, g_rot, g_scale = cot(g_w), cot(g_xyz) if blend else None, cot(g_rot) if blend else None, cot(g_scale) if blend else None
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_radius = new(m)
        g_weight = new(m) if keep["node_weight"] is not None else None
        g_trans, g_nrot, g_nscale = (new(m, 3), new(m, 4), new(m, 3)) if blend else (None, None, None)
        g_local = new(m, 4) if keep["node_local_rotation"] is not None else None
        lib = _lib()
        ws = torch.empty((lib.gsr_node_blend_workspace_size(n, m),), dtype=torch.uint8, device=dev)
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_backward(ctypes.byref(a), w.data_ptr(), dist.data_ptr(), idx.data_ptr(), p(g_xyz), p(g_rot), p(g_scale),
                                             p(g_w), p(g_trans), p(g_nrot), p(g_nscale), p(g_local), p(g_radius), p(g_weight), ws.data_ptr(),
                                             _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_backward")
        # inputs: x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation, K, residual, raw
        return None, None, None, g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local, None, None, None


class _NodeBlendBatch(torch.autograd.Function):
    """B blends of the same Gaussians and nodes with B sets of node attributes in one launch per stage (gsr_node_blend_*_batch): the
    views and flow partners of one mapping iteration. Inputs: x [n,3], motion_mask [n] | None, nodes [m,>=3], node_radius [m], node_weight
    [m] | None, node_trans [B,m,3], node_rot [B,m,4], node_scale [B,m,3], local_rotation [B,m,4] | None. Outputs d_xyz [B,n,3],
    d_rotation [B,n,4], d_scaling [B,n,3]."""

    @staticmethod
    def forward(ctx, x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation, K, rot_as_residual, raw):
        if not 1 <= K <= BLEND_MAX_K:
            raise ValueError(f"node blend: K = {K} outside 1..{BLEND_MAX_K}")
        x, nodes, node_radius = _f32(x, "x"), _f32(nodes, "nodes"), _f32(node_radius, "node_radius").reshape(-1)
        n, m, B = x.shape[0], nodes.shape[0], node_trans.shape[0]
        if x.dim() != 2 or x.shape[1] != 3 or nodes.dim() != 2 or nodes.shape[1] < 3 or node_trans.dim() != 3:
            raise ValueError(f"node blend batch expects x [N, 3], nodes [M, >=3], node_trans [B, M, 3]; got {tuple(x.shape)}, {tuple(nodes.shape)}, {tuple(node_trans.shape)}")
        opt = lambda t, name, shape: None if t is None else _checked(_f32(t, name), name, shape)
        motion_mask = opt(motion_mask, "motion_mask", None)
        if motion_mask is not None and motion_mask.numel() != n:
            raise ValueError(f"motion_mask must have one value per Gaussian ({n}), got {tuple(motion_mask.shape)}")
        node_weight = None if node_weight is None else _f32(node_weight, "node_weight").reshape(-1)
        if node_radius.numel() != m or (node_weight is not None and node_weight.numel() != m):
            raise ValueError("node_radius / node_weight must have one value per node")
        keep = dict(x=x, motion_mask=motion_mask, nodes=nodes, node_radius=node_radius, node_weight=node_weight,
                    node_trans=opt(node_trans, "node_trans", (B, m, 3)), node_rot=opt(node_rot, "node_rot", (B, m, 4)),
                    node_scale=opt(node_scale, "node_scale", (B, m, 3)), node_local_rotation=opt(local_rotation, "local_rotation", (B, m, 4)))
        scalars = dict(n=n, m=m, K=K, local_frame=int(local_rotation is not None), rot_as_residual=int(bool(rot_as_residual)),
                       node_stride=nodes.shape[1], flags=(RADIUS_IS_LOG | WEIGHT_IS_LOGIT) if raw else 0)
        a = _Blend(**scalars)
        for k, t in keep.items():
            setattr(a, k, t.data_ptr() if t is not None else None)
        dev = x.device
        w = torch.empty((n, K), dtype=torch.float32, device=dev)
        dist = torch.empty((n, K), dtype=torch.float32, device=dev)
        idx = torch.empty((n, K), dtype=torch.int64, device=dev)
        outs = [torch.empty((B, n, c), dtype=torch.float32, device=dev) for c in (3, 4, 3)]
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_forward_batch(ctypes.byref(a), B, w.data_ptr(), dist.data_ptr(), idx.data_ptr(), *(o.data_ptr() for o in outs), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_forward_batch")
        ctx.keep, ctx.scalars, ctx.saved, ctx.B = keep, scalars, (w, dist, idx), B
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scale):
        keep, sc, B = ctx.keep, ctx.scalars, ctx.B
        n, m = sc["n"], sc["m"]
        w, dist, idx = ctx.saved
        dev = w.device
        a = _Blend(**sc)
        for k, t in keep.items():
            setattr(a, k, t.data_ptr() if t is not None else None)
        cot = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_xyz, g_rot, g_scale = cot(g_xyz), cot(g_rot), cot(g_scale)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_radius = new(B, m)
        g_weight = new(B, m) if keep["node_weight"] is not None else None
        g_trans, g_nrot, g_nscale = new(B, m, 3), new(B, m, 4), new(B, m, 3)
        g_local = new(B, m, 4) if keep["node_local_rotation"] is not None else None
        lib = _lib()
        ws = torch.empty((lib.gsr_node_blend_workspace_size_batch(n, m, B),), dtype=torch.uint8, device=dev)
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_backward_batch(ctypes.byref(a), B, w.data_ptr(), dist.data_ptr(), idx.data_ptr(), p(g_xyz), p(g_rot), p(g_scale), None,
                                                   p(g_trans), p(g_nrot), p(g_nscale), p(g_local), p(g_radius), p(g_weight), ws.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_backward_batch")
        # radius / weight are shared by the B blends: their gradient is the sum of the rows
        g_radius = g_radius.sum(0).view(ctx.keep["node_radius"].shape)
        g_weight = None if g_weight is None else g_weight.sum(0)
        # inputs: x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation, K, residual, raw
        return None, None, None, g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local, None, None, None


def node_blend_batch(x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation=None, K: int = 3,
                     d_rot_as_res: bool = True, raw: bool = True):
    """node_blend for B sets of node attributes at once: node_trans [B, M, 3], node_rot [B, M, 4], node_scale [B, M, 3], local_rotation
    [B, M, 4] | None -> (d_xyz [B, N, 3], d_rotation [B, N, 4], d_scaling [B, N, 3]). Values and gradients are those of B node_blend calls
    (the radius / weight gradients are their sum)."""
    return _NodeBlendBatch.apply(x, motion_mask, nodes, node_radius.reshape(-1), _flat(node_weight), node_trans, node_rot, node_scale, local_rotation,
                                 K, d_rot_as_res, raw)


class _NodeBlendBatchPacked(torch.autograd.Function):
    """_NodeBlendBatch with the four node attributes as ONE matrix attrs [B, m, 14] = [d_xyz | d_rotation | d_scaling | local_rotation] -- the
    node network's heads as its single head layer produces them (gsr_node_blend.attr_stride / grad_stride): no per-attribute copies on the way
    in, one [B, m, 14] gradient on the way out (instead of four tensors that autograd concatenates)."""

    COLS = (0, 3, 7, 10, 14)              # column ranges of d_xyz, d_rotation, d_scaling, local_rotation

    @staticmethod
    def forward(ctx, x, motion_mask, nodes, node_radius, node_weight, attrs, K, rot_as_residual, raw):
        if not 1 <= K <= BLEND_MAX_K:
            raise ValueError(f"node blend: K = {K} outside 1..{BLEND_MAX_K}")
        x, nodes, node_radius = _f32(x, "x"), _f32(nodes, "nodes"), _f32(node_radius, "node_radius").reshape(-1)
        attrs = _f32(attrs, "attrs")
        n, m = x.shape[0], nodes.shape[0]
        if x.dim() != 2 or x.shape[1] != 3 or nodes.dim() != 2 or nodes.shape[1] < 3 or attrs.dim() != 3 or tuple(attrs.shape[1:]) != (m, 14):
            raise ValueError(f"node blend (packed) expects x [N, 3], nodes [M, >=3], attrs [B, M, 14]; got {tuple(x.shape)}, {tuple(nodes.shape)}, {tuple(attrs.shape)}")
        B = attrs.shape[0]
        motion_mask = None if motion_mask is None else _f32(motion_mask, "motion_mask")
        if motion_mask is not None and motion_mask.numel() != n:
            raise ValueError(f"motion_mask must have one value per Gaussian ({n}), got {tuple(motion_mask.shape)}")
        node_weight = None if node_weight is None else _f32(node_weight, "node_weight").reshape(-1)
        if node_radius.numel() != m or (node_weight is not None and node_weight.numel() != m):
            raise ValueError("node_radius / node_weight must have one value per node")
        keep = dict(x=x, motion_mask=motion_mask, nodes=nodes, node_radius=node_radius, node_weight=node_weight, attrs=attrs)
        scalars = dict(n=n, m=m, K=K, local_frame=1, rot_as_residual=int(bool(rot_as_residual)), node_stride=nodes.shape[1],
                       flags=(RADIUS_IS_LOG | WEIGHT_IS_LOGIT) if raw else 0, attr_stride=14, grad_stride=14)
        a = _NodeBlendBatchPacked._descriptor(scalars, keep)
        dev = x.device
        w = torch.empty((n, K), dtype=torch.float32, device=dev)
        dist = torch.empty((n, K), dtype=torch.float32, device=dev)
        idx = torch.empty((n, K), dtype=torch.int64, device=dev)
        outs = [torch.empty((B, n, c), dtype=torch.float32, device=dev) for c in (3, 4, 3)]
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_forward_batch(ctypes.byref(a), B, w.data_ptr(), dist.data_ptr(), idx.data_ptr(), *(o.data_ptr() for o in outs), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_forward_batch")
        ctx.keep, ctx.scalars, ctx.saved, ctx.B = keep, scalars, (w, dist, idx), B
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def _descriptor(scalars, keep):
        a = _Blend(**scalars)
        for k in ("x", "motion_mask", "nodes", "node_radius", "node_weight"):
            setattr(a, k, keep[k].data_ptr() if keep[k] is not None else None)
        base, c = keep["attrs"].data_ptr(), _NodeBlendBatchPacked.COLS
        a.node_trans, a.node_rot, a.node_scale, a.node_local_rotation = base + 4 * c[0], base + 4 * c[1], base + 4 * c[2], base + 4 * c[3]
        return a

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scale):
        keep, sc, B = ctx.keep, ctx.scalars, ctx.B
        n, m = sc["n"], sc["m"]
        w, dist, idx = ctx.saved
        dev = w.device
        a = _NodeBlendBatchPacked._descriptor(sc, keep)
        cot = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_xyz, g_rot, g_scale = cot(g_xyz), cot(g_rot), cot(g_scale)
        g_radius = torch.empty((B, m), dtype=torch.float32, device=dev)
        g_weight = torch.empty((B, m), dtype=torch.float32, device=dev) if keep["node_weight"] is not None else None
        g_attrs = torch.empty((B, m, 14), dtype=torch.float32, device=dev)
        lib = _lib()
        ws = torch.empty((lib.gsr_node_blend_workspace_size_batch(n, m, B),), dtype=torch.uint8, device=dev)
        p = lambda t: t.data_ptr() if t is not None else None
        base, c = g_attrs.data_ptr(), _NodeBlendBatchPacked.COLS
        with torch.cuda.device(dev):
            rc = lib.gsr_node_blend_backward_batch(ctypes.byref(a), B, w.data_ptr(), dist.data_ptr(), idx.data_ptr(), p(g_xyz), p(g_rot), p(g_scale), None,
                                                   base + 4 * c[0], base + 4 * c[1], base + 4 * c[2], base + 4 * c[3], p(g_radius), p(g_weight), ws.data_ptr(),
                                                   _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_node_blend_backward_batch")
        g_radius = g_radius.sum(0).view(keep["node_radius"].shape)
        g_weight = None if g_weight is None else g_weight.sum(0)
        # inputs: x, motion_mask, nodes, node_radius, node_weight, attrs, K, residual, raw
        return None, None, None, g_radius, g_weight, g_attrs, None, None, None


def node_blend_batch_packed(x, motion_mask, nodes, node_radius, node_weight, attrs, K: int = 3, d_rot_as_res: bool = True, raw: bool = True):
    """node_blend_batch (local frame) with the node attributes as one matrix attrs [B, M, 14] = [d_xyz | d_rotation | d_scaling |
    local_rotation]: same values and gradients, no copies of the four column ranges and one gradient matrix."""
    return _NodeBlendBatchPacked.apply(x, motion_mask, nodes, node_radius.reshape(-1), _flat(node_weight), attrs, K, d_rot_as_res, raw)


def _checked(t, name, shape):
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def cal_nn_weight(x, nodes, node_radius, node_weight=None, K: int = 3, raw: bool = True):
    """ControlNodeWarp.cal_nn_weight (:981-1011, gs_kernel=True).  raw=True (default): node_radius / node_weight are the module's RAW
    parameters _node_radius [M] / _node_weight [M, 1] and exp / sigmoid (:893-898) happen in the kernel; raw=False: they are the
    activated properties.  node_weight None: with_node_weight False.  Returns (nn_weight [N, K], nn_dist [N, K], nn_idx [N, K] int64)
    with gradients to node_radius / node_weight."""
    w, dist, idx, *_ = _NodeBlend.apply(x, None, nodes, node_radius.reshape(-1), _flat(node_weight), None, None, None, None, K, True, raw)
    return w, dist, idx


def _flat(node_weight):
    return None if node_weight is None else node_weight.reshape(-1)


def node_blend(x, motion_mask, nodes, node_radius, node_weight, node_trans, node_rot, node_scale, local_rotation=None, K: int = 3,
               d_rot_as_res: bool = True, raw: bool = True):
    """The body of ControlNodeWarp.forward (:1199-1258) after node_deform: blends the K nearest nodes' predictions
    node_trans = node_attrs['d_xyz'], node_rot = node_attrs['d_rotation'], node_scale = node_attrs['d_scaling'].
    local_rotation [M, 4] (node_attrs['local_rotation'], :1207) selects the local-frame translation; None = the global one.
    raw: see cal_nn_weight.  Returns {'d_xyz', 'd_rotation', 'd_scaling', 'nn_weight', 'nn_dist', 'nn_idx'}."""
    w, dist, idx, d_xyz, d_rot, d_scale = _NodeBlend.apply(x, motion_mask, nodes, node_radius.reshape(-1), _flat(node_weight), node_trans,
                                                           node_rot, node_scale, local_rotation, K, d_rot_as_res, raw)
    return {"d_xyz": d_xyz, "d_rotation": d_rot, "d_scaling": d_scale, "nn_weight": w, "nn_dist": dist, "nn_idx": idx}
