"""Control-node deformation model of the dynamic branch -- the part of the reference's SC-GS model (gaussian_splatting/scene/
deform_model.py DeformModel, utils/time_utils.py ControlNodeWarp :788-1300) that the SLAM loops call: ``deform.step(x, time_input,
...)`` -> {d_xyz, d_rotation, d_scaling, d_opacity, d_color}, ``deform.deform.expand_time``, ``arap_loss`` / ``elastic_loss``, an
``optimizer``. The per-Gaussian half (K nearest nodes, RBF weights, blend and all chain rules) is the HIP kernel set behind
``control_nodes.node_blend`` (include/control_nodes.h); the per-NODE half is a small time-conditioned MLP (O(512) rows, torch).

The network is the reference's DeformNetwork for the shipped flags (8 x 256, 10 + 10 frequencies, local frame); not built: hash-grid
encoders, node densification / pruning, hyper-coordinates, skinning -- the shipped SLAM configuration does not switch them on
(arguments.py:107-125) and the loops never call them."""

import functools
import os

import torch
from torch import nn

import control_nodes


def _embed(x, n_freq):
    """Positional encoding of utils/time_utils.py:208-273 (include_input, log-spaced sin / cos; order x, sin f0, cos f0, sin f1, ...):
    four launches whatever n_freq is."""
    if n_freq == 0:
        return x
    freqs = 2.0 ** torch.arange(n_freq, device=x.device, dtype=x.dtype)
    xf = x[..., None, :] * freqs[:, None]                                    # [..., F, C]
    sc = torch.stack([torch.sin(xf), torch.cos(xf)], -2)                      # [..., F, 2, C]
    return torch.cat([x, sc.flatten(-3)], -1)


def time_key(t):
    """Host-side name of a time sample: what ControlNodes.begin_iteration() files its batched evaluations under."""
    return round(float(t), 7)


def farthest_point_sample(xyz, npoint):
    """utils/time_utils.py:478-500 on one point set [N,3] -> indices [npoint] (deterministic start at index 0)."""
    N = xyz.shape[0]
    npoint = min(npoint, N)
    idx = torch.zeros(npoint, dtype=torch.long, device=xyz.device)
    dist = torch.full((N,), 1e10, device=xyz.device)
    far = torch.zeros((), dtype=torch.long, device=xyz.device)
    for i in range(npoint):
        idx[i] = far
        d = ((xyz - xyz[far]) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = torch.argmax(dist)
    return idx


# ---- the node regularisers of SC-GS as the reference runs them (utils/deform_utils.py, utils/time_utils.py:1128-1165) -----------------
def kabsch_rotations(S):
    """R = V U^T of S = U Sigma V^T with the reflection rule of estimate_rotation (deform_utils.py:152-162), for [..., 3, 3] matrices:
    one HIP launch (gsr_kabsch_rotations, include/slam_map.h) instead of torch.svd on thousands of 3x3 matrices."""
    from . import _lib
    Sc = S.detach().to(torch.float32).contiguous()
    R = torch.empty_like(Sc)
    n = Sc.numel() // 9
    if n:
        with torch.cuda.device(Sc.device):
            _lib.check(_lib.lib().gsr_kabsch_rotations(n, _lib.dev_f32(Sc, "S"), R.data_ptr(), _lib.stream(Sc.device)), "gsr_kabsch_rotations")
    return R


def connectivity_from_points(points, radius=0.1, K=10, least_edge_num=3):
    """cal_connectivity_from_points (deform_utils.py:58-110, mode 'nn'): the K nearest other points of every point; beyond the first
    `least_edge_num`, neighbours farther than `radius` are dropped. points [..., Nv, 3] -> (nn_idx [..., Nv, K] int64, keep [..., Nv, K] bool).
    (The edge weights that function also returns are not used by arap_loss: cal_arap_error is called without them, :1139-1140.)"""
    pts = points.detach().reshape(-1, points.shape[-2], 3)
    K = min(K, pts.shape[1] - 1)
    knn = control_nodes.knn_points(pts, pts, K=K + 1)
    nn_dist, nn_idx = knn.dists[:, :, 1:], knn.idx[:, :, 1:]
    # keep[..., :least_edge_num] = True, beyond that dist < radius^2: ONE comparison against a per-column bound (inf for the first columns; the
    # fill + compare + slice copy of the literal form were three launches per call)
    keep = nn_dist < _edge_bounds(K, least_edge_num, radius, nn_dist.device)
    shape = points.shape[:-1] + (K,)
    return nn_idx.reshape(shape), keep.reshape(shape)


_EDGE_BOUNDS = {}


def _edge_bounds(K, least_edge_num, radius, device):
    key = (K, least_edge_num, float(radius), str(device))
    t = _EDGE_BOUNDS.get(key)
    if t is None:
        if len(_EDGE_BOUNDS) >= 16:
            _EDGE_BOUNDS.pop(next(iter(_EDGE_BOUNDS)))
        # built ON the device (a host-to-device copy would break a stream capture that meets a new layout first)
        t = torch.full((K,), float(radius) ** 2, dtype=torch.float32, device=device)
        t[:least_edge_num] = float("inf")
        _EDGE_BOUNDS[key] = t
    return t


def edge_matrix(verts, nn_idx, keep):
    """produce_edge_matrix_nfmt (deform_utils.py:35-42): E[i, n] = verts[i] - verts[nn_idx[i, n]] on the kept edges, 0 elsewhere.
    verts [..., Nv, 3] with nn_idx / keep [..., Nv, K] (leading dimensions broadcast against those of verts)."""
    lead = verts.shape[:-2]
    Nv, K = nn_idx.shape[-2:]
    flat = verts.reshape(-1, verts.shape[-2], 3)
    if flat.is_cuda and flat.dtype == torch.float32:
        # device tensors: control_nodes.gather_rows -- the same values, and a backward pass that adds a vertex's incoming edge gradients in a
        # FIXED order (gsr_index_csr + gsr_segment_sum) where torch.gather's backward is a scatter_add with float atomics (not reproducible)
        sets_idx = nn_idx.reshape(-1, Nv * K)                                     # one index set per distinct leading index of nn_idx
        S, B = sets_idx.shape[0], flat.shape[0]
        sets = control_nodes.IndexSets(sets_idx, Nv)
        set_of_b = None
        if S > 1 and S != B:                                                      # nn_idx broadcasts over trailing leading axes of verts (the time samples)
            set_of_b = torch.arange(S, device=flat.device, dtype=torch.int32).repeat_interleave(B // S)
        nb = control_nodes.gather_rows(flat, sets, set_of_b).reshape(*lead, Nv, K, 3)
        return (verts[..., :, None, :] - nb) * keep[..., None]
    # a gather along the vertex axis of the [B, Nv, 3] table itself: its backward scatters into a table of that size (gathering from a
    # view expanded to [B, Nv, Nv, 3] would zero-fill, scatter into and reduce 3 MB per time sample at 512 nodes)
    idx = nn_idx.expand(*lead, Nv, K).reshape(-1, Nv * K, 1).expand(-1, -1, 3)
    nb = torch.gather(flat, 1, idx).reshape(*lead, Nv, K, 3)
    return (verts[..., :, None, :] - nb) * keep[..., None]


class _ArapTerm(torch.autograd.Function):
    """partial[v, t-1, m] = sum_k keep |E_t - R E_0|^2 of arap_error in one launch each way (gsr_arap_forward / _backward, include/slam_map.h):
    p [V, T, M, 3], nb [V, T, M, K, 3] (the gathered neighbours), keep [V, M, K] float."""

    @staticmethod
    def forward(ctx, p, nb, keep):
        from . import _lib
        V, T, M, K = (int(v) for v in nb.shape[:4])
        p, nb, keep = p.contiguous(), nb.contiguous(), keep.contiguous()
        R = torch.empty((V, T - 1, M, 9), dtype=torch.float32, device=p.device)
        partial = torch.empty((V, T - 1, M), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().gsr_arap_forward(V, T, M, K, p.data_ptr(), nb.data_ptr(), keep.data_ptr(), R.data_ptr(), partial.data_ptr(),
                                                   _lib.stream(p.device)), "gsr_arap_forward")
        ctx.save_for_backward(p, nb, keep, R)
        return partial

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        p, nb, keep, R = ctx.saved_tensors
        V, T, M, K = (int(v) for v in nb.shape[:4])
        g = g.to(torch.float32).contiguous()
        dp, dnb = torch.empty_like(p), torch.empty_like(nb)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().gsr_arap_backward(V, T, M, K, p.data_ptr(), nb.data_ptr(), keep.data_ptr(), R.data_ptr(), g.data_ptr(), dp.data_ptr(),
                                                    dnb.data_ptr(), _lib.stream(p.device)), "gsr_arap_backward")
        return dp, dnb, None


class _ElasticRatio(torch.autograd.Function):
    """ratio[v, m, k] = var_t |nb - x| / (its detached value + 1e-5) of elastic_error in one launch forward, two back (gsr_elastic_forward /
    _backward): x [V, M, T, 3], nb [V, M, K, T, 3]."""

    @staticmethod
    def forward(ctx, x, nb):
        from . import _lib
        V, M, K, T = (int(v) for v in nb.shape[:4])
        x, nb = x.contiguous(), nb.contiguous()
        ratio = torch.empty((V, M, K), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().gsr_elastic_forward(V, M, K, T, x.data_ptr(), nb.data_ptr(), ratio.data_ptr(), _lib.stream(x.device)), "gsr_elastic_forward")
        ctx.save_for_backward(x, nb)
        return ratio

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x, nb = ctx.saved_tensors
        V, M, K, T = (int(v) for v in nb.shape[:4])
        g = g.to(torch.float32).contiguous()
        dx, dnb = torch.empty_like(x), torch.empty_like(nb)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().gsr_elastic_backward(V, M, K, T, x.data_ptr(), nb.data_ptr(), g.data_ptr(), dx.data_ptr(), dnb.data_ptr(),
                                                       _lib.stream(x.device)), "gsr_elastic_backward")
        return dx, dnb


FUSED_REGULARISERS = os.environ.get("GSR_FUSED_REGULARISERS", "1") != "0"


def estimate_rotation(E0, Et, weight, rotations=kabsch_rotations):
    """estimate_rotation (deform_utils.py:130-166) on edge matrices [..., Nv, K, 3]: S = E0^T diag(w) Et, zeroed for vertices none of whose
    edges changed in some coordinate (:147-149), then the best-fit rotation of every S."""
    S = torch.einsum("...ka,...k,...kb->...ab", E0, weight, Et)
    unchanged = (E0 == Et).all(dim=-2).any(dim=-1)
    S = torch.where(unchanged[..., None, None], torch.zeros_like(S), S)
    return rotations(S)


_SET_OF_B = {}


def _set_of_b(V, T, offset, device):
    """[offset, ..., offset + V - 1] each repeated T times (int32, on the device): a constant of the layout, built once (arange +
    repeat_interleave were three launches per call)."""
    key = (V, T, offset, str(device))
    t = _SET_OF_B.get(key)
    if t is None:
        if len(_SET_OF_B) >= 64:                 # (a handful of layouts per run; bounded all the same)
            _SET_OF_B.pop(next(iter(_SET_OF_B)))
        # built ON the device: a host-to-device copy would break a stream capture that hits a new layout first
        t = _SET_OF_B[key] = (torch.arange(V, dtype=torch.int32, device=device) + offset).repeat_interleave(T)
        t._gsr_long = t.long()                   # (what gather_rows' index_select wants: no conversion launch per call)
    return t


def arap_error(nodes_seq, nn_idx, keep, rotations=kabsch_rotations, sets=None, set_offset=0):
    """cal_arap_error (deform_utils.py:177-205) without edge weights (every kept edge weighs 1) and without its random vertex subsample,
    which only starts above 512 vertices (the node budget). nodes_seq [..., T, Nv, 3]; nn_idx / keep [..., Nv, K] from
    connectivity_from_points(nodes_seq[..., 0, :, :]). Returns the error per leading index. The reference loops over the samples
    1 .. T-1 (:190-204); here they are one more tensor axis (same terms, summed in one reduction)."""
    if nodes_seq.shape[-3] < 2:
        return torch.zeros(nodes_seq.shape[:-3], dtype=nodes_seq.dtype, device=nodes_seq.device)
    if (FUSED_REGULARISERS and rotations is kabsch_rotations and nodes_seq.is_cuda and nodes_seq.dtype == torch.float32 and nodes_seq.dim() == 4
            and nn_idx.dim() == 3 and nn_idx.shape[0] == nodes_seq.shape[0]):
        # device tensors [V, T, M, 3] with one neighbour set per view: the neighbours through control_nodes.gather_rows (ordered backward),
        # everything per (view, sample, node) -- edges, cross-covariance, Kabsch, the error and all of its backward -- in one launch each way
        V, T, M, _ = nodes_seq.shape
        K = nn_idx.shape[-1]
        if sets is None:              # (`sets`: the reverse lists of a larger family of neighbour sets, of which nn_idx is the rows from set_offset)
            sets, set_offset = control_nodes.IndexSets(nn_idx.reshape(V, M * K), M), 0
        set_of_b = _set_of_b(V, T, set_offset, nodes_seq.device) if sets.S > 1 else None
        nb = control_nodes.gather_rows(nodes_seq.reshape(V * T, M, 3), sets, set_of_b).reshape(V, T, M, K, 3)
        return _ArapTerm.apply(nodes_seq, nb, keep.to(torch.float32)).sum(dim=(1, 2))
    w = keep.to(nodes_seq.dtype)
    E = edge_matrix(nodes_seq, nn_idx[..., None, :, :], keep[..., None, :, :])          # [..., T, Nv, K, 3]
    E0, Et = E.split([1, E.shape[-4] - 1], -4)
    with torch.no_grad():
        R = estimate_rotation(E0.detach(), Et.detach(), w[..., None, :, :], rotations)  # [..., T-1, Nv, 3, 3]
    stretch = Et - torch.einsum("...ab,...kb->...ka", R, E0)                            # target edges minus the rigidly rotated source edges
    return (w[..., None, :, :] * stretch.square().sum(dim=-1)).sum(dim=(-3, -2, -1))


def elastic_error(nodes_t, nn_weight, nn_idx):
    """The body of ControlNodeWarp.elastic_loss (time_utils.py:1160-1165): the variance over time of every edge length to the K nearest
    nodes, normalised by its own detached value, weighted by the RBF weights. nodes_t [..., M, T, 3]; nn_weight / nn_idx [M, K]."""
    if nodes_t.is_cuda and nodes_t.dtype == torch.float32:
        # (device tensors: the neighbour rows through control_nodes.gather_rows -- advanced indexing's backward is an index_put with float atomics)
        M, K = nn_idx.shape
        T = nodes_t.shape[-2]
        lead = nodes_t.shape[:-3]
        flat = nodes_t.reshape(-1, M, T * 3)
        nb = control_nodes.gather_rows(flat, control_nodes.IndexSets(nn_idx.reshape(1, M * K), M)).reshape(*lead, M, K, T, 3)
        if FUSED_REGULARISERS and len(lead) == 1 and 2 <= T <= 16:
            ratio = _ElasticRatio.apply(flat.reshape(lead[0], M, T, 3), nb)            # [V, M, K]: the variance and its normalisation, fused
            return (ratio * nn_weight).sum(dim=-1).mean(dim=-1)
        edge_t = (nb - nodes_t[..., :, None, :, :]).norm(dim=-1)                          # [..., M, K, T]
    else:
        edge_t = (nodes_t[..., nn_idx, :, :] - nodes_t[..., :, None, :, :]).norm(dim=-1)      # [..., M, K, T]
    var = edge_t.var(dim=-1)
    var = var / (var.detach() + 1e-5)
    return (var * nn_weight).sum(dim=-1).mean(dim=-1)


def draw_loss_times(t, arap_delta, arap_samples, elastic_delta, elastic_samples=8):
    """The time samples ControlNodeWarp.arap_loss (:1129-1132) and elastic_loss (:1145-1149) draw around t, as host floats and in the
    reference's order of torch.rand calls (CPU generator here, the reference draws on the device)."""
    ta = t + arap_delta * (float(torch.rand(())) - 0.5)
    arap = (torch.rand(arap_samples) * arap_delta + ta - 0.5 * arap_delta).tolist()
    te = t + elastic_delta * (float(torch.rand(())) - 0.5)
    elastic = (torch.rand(elastic_samples) * elastic_delta + te - 0.5 * elastic_delta).tolist()
    return {"arap": arap, "elastic": elastic}


def draw_arap_times(t, delta, samples):
    """The time samples ONE ControlNodeWarp.arap_loss(t, delta_t, t_samp_num) call draws (:1129-1132), as host floats (CPU generator)."""
    ta = t + delta * (float(torch.rand(())) - 0.5)
    return (torch.rand(samples) * delta + ta - 0.5 * delta).tolist()


@functools.lru_cache(maxsize=256)
def _row_groups(R, target=2048):
    """Into how many equal parts to split R rows so that a part has about `target` rows: the divisor g of R with R / g closest to the target
    (within a factor of two of it), else 1."""
    best, best_err = 1, None
    for g in range(max(1, R // (2 * target)), R // (target // 2) + 2):
        if g > 0 and R % g == 0:
            err = abs(R // g - target)
            if best_err is None or err < best_err:
                best, best_err = g, err
    return best


def _grad_weight(G, X):
    """dW = G^T X for G [R, out], X [R, in] with the rows split into equal parts of ~2 000: ONE batched GEMM + a sum over the parts. As a
    single GEMM the library tiles the [out, in] = 256 x 256 result into 32 workgroups and walks all R rows in each -- 224 of the 256 CUs
    idle, 35-40 TFLOP/s (118 us per layer at R = 33 280); as 13-26 independent products of 1 280-2 560 rows it runs at 81-83 TFLOP/s (52 us;
    tools/dev_dw_gemm.py: parts of 512 rows or fewer are slow again, the library then picks one 256 x 256 tile per part). The parts are added
    in a fixed order."""
    R = G.shape[0]
    groups = _row_groups(R)
    if groups > 1 and G.is_contiguous() and X.is_contiguous():
        return torch.bmm(G.view(groups, R // groups, -1).transpose(1, 2), X.view(groups, R // groups, -1)).sum(0)
    return G.t().mm(X)


LAYER_FUSED_TRUNK = os.environ.get("GSR_LAYER_FUSED_TRUNK", "0") == "1"
# The trunk's forward and input-gradient products on the bf16 matrix cores with fp32-accurate three-term operands (include/dense_layers.h,
# dense_layers.py) instead of the library's fp32 GEMMs: fp32-GEMM accuracy, ~20 % less time per product at the SLAM runs' ~33 k rows, and the
# backward's ReLU mask + bias gradient ride in the input-gradient product's epilogue (no separate pass over the rows). GSR_DENSE_TRUNK=0: the
# library path.
DENSE_TRUNK = os.environ.get("GSR_DENSE_TRUNK", "1") == "1"
# ... and the eight layers (and the seven input-gradient products on the way back) as ONE launch each (gsr_dense_chain: a block carries its rows
# through all layers). GSR_DENSE_CHAIN=0: one launch per product.
DENSE_CHAIN = os.environ.get("GSR_DENSE_CHAIN", "1") == "1"
# ... and the weight gradients of all layers + heads, which the chained backward leaves as independent products over the same rows, as ONE launch
# on the same three-term operands (gsr_dense_wgrad_many). GSR_DENSE_WGRAD_MANY=0: the library's batched fp32 GEMMs (rounds 4-5).
DENSE_WGRAD_MANY = os.environ.get("GSR_DENSE_WGRAD_MANY", "1") == "1"


def _dense_trunk_ok(emb, Ws, W_heads, skip):
    Wd, E = int(Ws[0].shape[0]), int(emb.shape[1])
    return (DENSE_TRUNK and Wd % 128 == 0 and E % 4 == 0 and emb.is_contiguous() and len(Ws) <= 12
            and all(int(w.shape[0]) == Wd and w.is_contiguous() for w in Ws)
            and all(int(Ws[i].shape[1]) == (Wd + E if i == skip + 1 else Wd) for i in range(1, len(Ws))) and int(Ws[0].shape[1]) == E)


class _FusedTrunk(torch.autograd.Function):
    """The node network on the device as ONE autograd node: D layers y = relu(x W^T + b) with the embedding re-injected behind layer `skip`
    (utils/time_utils.py:428-452), then all heads as one linear layer [sum of head widths, W] without activation. What autograd's op-by-op
    version spends besides the forward / input-gradient GEMMs on ~30-70 000 rows x 256 columns is folded away or re-shaped here:
      * forward: bias + ReLU ride in the GEMM's epilogue (torch._addmm_activation -> hipBLASLt RELU_BIAS);
      * backward: G = dY [y > 0] and the bias gradient in one pass (control_nodes.relu_backward_bias, fixed summation order);
      * the weight gradients G^T x -- as single GEMMs the slowest kernels of the iteration (256 x 256 results: 32 workgroups) -- are batched
        over row groups (_grad_weight);
      * the skip connection's concatenation is kept (one GEMM on [emb | h]), its backward half is not: the gradient of the concatenated
        input is only formed for its h columns (G W[:, E:]); the embedding needs no gradient (node positions are detached, times are data).
    Values are those of the op-by-op network; gradients differ from it by summation order only."""

    @staticmethod
    def forward(ctx, emb, skip, W_heads, b_heads, *params):
        D = len(params) // 2
        E = emb.shape[1]
        ctx.planes_t = ctx.heads_planes_t = None
        if LAYER_FUSED_TRUNK:
            # opt-in (GSR_LAYER_FUSED_TRUNK=1): the eight layers + heads as ONE launch on the bf16 matrix cores with fp32-accurate three-term
            # operands (include/dense_layers.h gsr_trunk_forward). Same values to fp32-GEMM accuracy; faster than the library from ~50 k rows
            # (696 vs 875 us at 66 k), slower at the SLAM runs' ~33 k (420 vs 389 us): off by default. The backward pass below is unchanged.
            import dense_layers
            Ws, bs = list(params[0::2]), list(params[1::2])
            if dense_layers.trunk_supported(emb, Ws, skip, W_heads):
                out, inputs, outs = dense_layers.trunk_forward(emb, [w.detach() for w in Ws], [b.detach() for b in bs], W_heads.detach(), b_heads.detach())
                ctx.skip, ctx.D, ctx.E = skip, D, E
                ctx.save_for_backward(W_heads, *params[0::2], *inputs, *outs)
                return out
        Ws = list(params[0::2])
        if _dense_trunk_ok(emb, Ws, W_heads, skip):
            import dense_layers
            Wd, R = int(Ws[0].shape[0]), int(emb.shape[0])
            # every layer's weight as bf16 planes, and the transposed planes of its hidden columns for the way back: one launch
            requests = [(w.detach(), 0, None, False) for w in Ws] + [(Ws[i].detach(), E if i == skip + 1 else 0, Wd, True) for i in range(1, D)]
            heads_dense = DENSE_WGRAD_MANY and int(W_heads.shape[1]) == Wd and W_heads.is_contiguous() and len(requests) + 2 <= 24
            if heads_dense:                                  # the heads' weight too, both orientations (no library GEMM left in the network)
                requests += [(W_heads.detach(), 0, None, False), (W_heads.detach(), 0, Wd, True)]
            views, ctx.planes_buffer = dense_layers.split_weights(requests)
            cat = emb.new_empty((R, E + Wd))                 # (:447-448: [emb | h], the input of layer skip + 1; layer skip writes its half)
            cat[:, :E] = emb
            inputs, outs, h = [], [], emb
            chained = DENSE_CHAIN and Wd == dense_layers.CHAIN_WIDTH and D <= dense_layers.CHAIN_MAX
            ops = []
            for i in range(D):
                inputs.append(h)
                y = cat[:, E:] if i == skip else emb.new_empty((R, Wd))
                if chained:
                    ops.append(dict(X=h, planes=views[i], K=int(Ws[i].shape[1]), bias=params[2 * i + 1].detach(), relu=True, Y=y))
                else:
                    dense_layers.dense_forward(h, views[i], Wd, int(Ws[i].shape[1]), params[2 * i + 1].detach(), relu=True, out=y)
                outs.append(y)
                h = cat if i == skip else y
            if chained:
                dense_layers.dense_chain(ops)
            if heads_dense and h.is_contiguous():
                out = dense_layers.dense_forward(h, views[2 * D - 1], int(W_heads.shape[0]), Wd, b_heads.detach())
            else:
                out = torch.addmm(b_heads, h, W_heads.t())
            ctx.planes_t = views[D:2 * D - 1]
            ctx.heads_planes_t = views[2 * D] if heads_dense else None
            ctx.skip, ctx.D, ctx.E = skip, D, E
            ctx.save_for_backward(W_heads, *params[0::2], *inputs, *outs)
            return out
        inputs, outs, h = [], [], emb
        for i in range(D):
            W, b = params[2 * i], params[2 * i + 1]
            inputs.append(h)
            h = torch._addmm_activation(b, h, W.t(), use_gelu=False)
            outs.append(h)
            if i == skip:
                h = torch.cat([emb, h], -1)                  # (:447-448: the next layer's input)
        out = torch.addmm(b_heads, h, W_heads.t())
        ctx.skip, ctx.D, ctx.E = skip, D, E
        ctx.save_for_backward(W_heads, *params[0::2], *inputs, *outs)
        return out

    @staticmethod
    def backward(ctx, g_out):
        D, E, skip = ctx.D, ctx.E, ctx.skip
        saved = ctx.saved_tensors
        W_heads, Ws, inputs, outs = saved[0], saved[1:1 + D], saved[1 + D:1 + 2 * D], saved[1 + 2 * D:]
        grads = [None] * (2 * D)
        g_out = g_out.contiguous()
        h_last = outs[D - 1]
        wgrad_many = (DENSE_WGRAD_MANY and ctx.planes_t is not None and DENSE_CHAIN and D - 1 <= 8 and int(Ws[0].shape[0]) == 256
                      and D + 1 <= 12 and h_last.is_contiguous())
        gW_heads, gb_heads = (None if wgrad_many else _grad_weight(g_out, h_last)), g_out.sum(0)
        heads_t = getattr(ctx, "heads_planes_t", None) if wgrad_many else None
        g = None if heads_t is not None else g_out.mm(W_heads)
        # the weight gradients of the layers of one shape (six of the eight are [W, W]) are batched products into ONE buffer, summed over their
        # row groups by one launch at the end instead of one per layer (they feed nothing on the way back)
        R = int(g_out.shape[0])
        groups = _row_groups(R)
        same = [i for i in range(D) if tuple(Ws[i].shape) == tuple(Ws[D - 1].shape) and inputs[i].is_contiguous()] if groups > 1 else []
        buf = g_out.new_empty((len(same), groups) + tuple(Ws[D - 1].shape)) if len(same) > 1 else None
        planes_t = ctx.planes_t
        G = db = None
        chain_G = None
        if planes_t is not None and DENSE_CHAIN and D - 1 <= 8 and int(Ws[0].shape[0]) == 256:
            # all input-gradient products first, as ONE launch (each hands its result to the ReLU of the layer below and sums that layer's bias
            # gradient), then the weight gradients from the G's they left
            import dense_layers
            Wd = int(Ws[0].shape[0])
            if heads_t is not None:      # (g_out W_heads) [y > 0] and its column sums in one pass: the last layer's G and bias gradient
                G, db = dense_layers.dense_backward_input(g_out, heads_t, Wd, int(W_heads.shape[0]), mask=outs[D - 1])
            else:
                G, db = control_nodes.relu_backward_bias(g, outs[D - 1])
            chain_G, chain_db = [None] * D, [None] * D
            chain_G[D - 1], chain_db[D - 1] = G, db
            ops = []
            for i in range(D - 1, 0, -1):
                chain_G[i - 1], chain_db[i - 1] = G.new_empty((R, Wd)), G.new_empty((Wd,))
                ops.append(dict(X=chain_G[i], planes=planes_t[i - 1], K=Wd, Y=chain_G[i - 1], mask=outs[i - 1], dbias=chain_db[i - 1]))
            dense_layers.dense_chain(ops)
            if wgrad_many:
                # every layer's G and input now exist: the D + 1 weight gradients as ONE launch (gsr_dense_wgrad_many), no library GEMM
                dWs = dense_layers.dense_wgrad_many([(chain_G[i], inputs[i]) for i in range(D)] + [(g_out, h_last)])
                for i in range(D):
                    grads[2 * i], grads[2 * i + 1] = dWs[i], chain_db[i]
                return (None, None, dWs[D], gb_heads, *grads)
        for i in reversed(range(D)):
            if chain_G is not None:
                G, db = chain_G[i], chain_db[i]
            elif planes_t is None or i == D - 1:
                G, db = control_nodes.relu_backward_bias(g, outs[i])
            if buf is not None and i in same:
                torch.bmm(G.view(groups, R // groups, -1).transpose(1, 2), inputs[i].view(groups, R // groups, -1), out=buf[same.index(i)])
            else:
                grads[2 * i] = _grad_weight(G, inputs[i])
            grads[2 * i + 1] = db
            if chain_G is not None:
                continue
            if i > 0 and planes_t is not None:
                # the layer below's G and bias gradient straight from this layer's input-gradient product (mask and column sums in its epilogue)
                import dense_layers
                Wd = int(Ws[i].shape[0])
                G, db = dense_layers.dense_backward_input(G, planes_t[i - 1], Wd, Wd, mask=outs[i - 1])
            elif i > 0:
                g = G.mm(Ws[i][:, E:] if i == skip + 1 else Ws[i])      # (the embedding half of the skip input needs no gradient)
        if buf is not None:
            for j, dW in enumerate(buf.sum(1).unbind(0)):
                grads[2 * same[j]] = dW
        return (None, None, gW_heads, gb_heads, *grads)


class NodeNetwork(nn.Module):
    """DeformNetwork (utils/time_utils.py:327-470) as ControlNodeWarp builds it with the shipped flags (arguments.py:107-125: is_blender
    False -> 10 time frequencies, 10 position frequencies, D = 8 layers of W = 256 with the embedding re-injected after layer 4,
    local_frame True, no opacity / colour heads). Same attribute names as the reference, so its state_dict loads."""

    def __init__(self, D=8, W=256, multires=10, t_multires=10, local_frame=True):
        super().__init__()
        self.D, self.W, self.multires, self.t_multires, self.local_frame = D, W, multires, t_multires, local_frame
        self.skips = [D // 2]
        self.input_ch = 3 * (1 + 2 * multires) + (1 + 2 * t_multires)
        self.linear = nn.ModuleList([nn.Linear(self.input_ch, W)] +
                                    [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.gaussian_warp, self.gaussian_scaling, self.gaussian_rotation = nn.Linear(W, 3), nn.Linear(W, 3), nn.Linear(W, 4)
        if local_frame:
            self.local_rotation = nn.Linear(W, 4)
            nn.init.normal_(self.local_rotation.weight, mean=0, std=1e-4)
            nn.init.zeros_(self.local_rotation.bias)
        for layer in self.linear:                                                    # :390-392
            nn.init.kaiming_uniform_(layer.weight, mode="fan_in", nonlinearity="relu")
            nn.init.zeros_(layer.bias)
        nn.init.normal_(self.gaussian_warp.weight, mean=0, std=1e-5)                 # :394-399: (almost) the identity deformation at start
        nn.init.normal_(self.gaussian_scaling.weight, mean=0, std=1e-8)
        nn.init.normal_(self.gaussian_rotation.weight, mean=0, std=1e-5)
        for head in (self.gaussian_warp, self.gaussian_scaling, self.gaussian_rotation):
            nn.init.zeros_(head.bias)

    def forward(self, x, t):
        """x [N,3], t [N,1] -> {d_xyz, d_rotation, d_scaling, local_rotation} (:428-470)."""
        emb = torch.cat([_embed(x, self.multires), _embed(t, self.t_multires)], -1)
        return self.from_embedding(emb)

    def fused_ok(self, emb):
        return (emb.is_cuda and emb.dtype == torch.float32 and emb.dim() == 2 and self.W in control_nodes.RELU_BIAS_COLS and len(self.skips) == 1
                and 0 <= self.skips[0] < self.D - 1 and not emb.requires_grad and os.environ.get("GSR_FUSED_TRUNK", "1") != "0")

    def heads(self):
        """[(output name, layer)] in the column order of heads_from_embedding."""
        hs = [("d_xyz", self.gaussian_warp), ("d_rotation", self.gaussian_rotation), ("d_scaling", self.gaussian_scaling)]
        if self.local_frame:
            hs.append(("local_rotation", self.local_rotation))
        return hs

    def heads_from_embedding(self, emb):
        """All heads of all rows of `emb` as ONE matrix [rows, 3 + 4 + 3 (+ 4)] (columns in heads() order): the heads as one linear layer on
        the trunk's output -- four GEMMs forward and eight backward become one and two, and the trunk's output gradient is formed once.
        On the device the whole network is one autograd node (_FusedTrunk)."""
        hs = self.heads()
        W_all = torch.cat([m.weight for _, m in hs], 0)
        b_all = torch.cat([m.bias for _, m in hs], 0)
        if self.fused_ok(emb):
            params = [t for layer in self.linear for t in (layer.weight, layer.bias)]
            return _FusedTrunk.apply(emb, self.skips[0], W_all, b_all, *params)
        return torch.addmm(b_all, self.trunk(emb), W_all.t())

    def trunk(self, emb):
        h = emb
        for i, layer in enumerate(self.linear):
            h = torch.relu(layer(h))
            if i in self.skips:
                h = torch.cat([emb, h], -1)
        return h

    def from_embedding(self, emb):
        h = self.trunk(emb)
        out = {"d_xyz": self.gaussian_warp(h), "d_rotation": self.gaussian_rotation(h), "d_scaling": self.gaussian_scaling(h), "d_opacity": None,
               "d_color": None}
        if self.local_frame:
            out["local_rotation"] = self.local_rotation(h)
        return out


class ControlNodes(nn.Module):
    """ControlNodeWarp (utils/time_utils.py:788-1300) for the shipped flags: K = 3 nearest of up to 512 nodes, RBF weights with learnt
    radius and node weight, local-frame translation, residual rotation."""

    def __init__(self, node_num=512, K=3, d_rot_as_res=True, local_frame=True, device="cuda", D=8, W=256):
        super().__init__()
        self.K, self.max_nodes, self.d_rot_as_res, self.local_frame = K, node_num, d_rot_as_res, local_frame
        self.device = torch.device(device)
        self.network = NodeNetwork(D=D, W=W, local_frame=local_frame).to(self.device)
        self.nodes = nn.Parameter(torch.zeros(0, 3, device=self.device))
        self._node_radius = nn.Parameter(torch.zeros(0, device=self.device))
        self._node_weight = nn.Parameter(torch.zeros(0, 1, device=self.device))
        self.reg_loss = 0.0
        self.inited = False
        self._batch = None            # {time_key: network outputs [M, .]} of the current iteration (begin_iteration)
        self._blended = None          # the Gaussians' deltas at the iteration's full samples, from one batched blend (begin_iteration(blend=...))
        self._graph = None

    node_num = property(lambda s: s.nodes.shape[0])

    def trainable_parameters(self):
        """:844-853 (with_node_weight): the network and the node tensors as two groups."""
        return [{"params": list(self.network.parameters()), "name": "deform"},
                {"params": [self.nodes, self._node_radius, self._node_weight], "name": "nodes"}]

    @staticmethod
    def _initial_radius(init_pcl, n, device):
        """log(0.1 * scene range + 1e-7) for every node, scene range = max - min over ALL coordinates of the seeding points (:929,:941-945)."""
        scene_range = init_pcl.max() - init_pcl.min()
        return torch.log(0.1 * scene_range + 1e-7) * torch.ones(n, dtype=torch.float32, device=device)

    @torch.no_grad()
    def init(self, init_pcl, **_):
        """ControlNodeWarp.init (:904-951): every point becomes a node while there are fewer points than the node budget, else
        farthest-point sampling; radius from the scene range, node weights 0."""
        pts = init_pcl.detach().float()
        if self.max_nodes > pts.shape[0]:
            nodes = pts.clone()
        else:
            nodes = pts[farthest_point_sample(pts, self.max_nodes)].clone()
        self.nodes = nn.Parameter(nodes)
        self._node_radius = nn.Parameter(self._initial_radius(pts, nodes.shape[0], self.device))
        self._node_weight = nn.Parameter(torch.zeros(nodes.shape[0], 1, device=self.device))
        self.inited = True

    @torch.no_grad()
    def extend_node(self, init_pcl, sample_number=250, **_):
        """extend_node (:953-981) + DeformModel.extend_node_from_point (deform_model.py:71-95): new dynamic points add nodes -- all of them
        while they are fewer than the current node count, else `sample_number` by farthest-point sampling (that branch of the reference
        dereferences an unset variable; the radius rule of the other branch is used for it here). Returns the new (nodes, radius, weight)
        rows; the caller appends them and extends the optimizer state."""
        pts = init_pcl.detach().float()
        if pts.shape[0] == 0:
            return None
        new_nodes = pts.clone() if self.node_num > pts.shape[0] else pts[farthest_point_sample(pts, sample_number)].clone()
        return new_nodes, self._initial_radius(pts, new_nodes.shape[0], self.device), torch.zeros(new_nodes.shape[0], 1, device=self.device)

    def expand_time(self, t):
        """:975-979."""
        return t.reshape(1, 1).expand(self.node_num, 1)

    # ---- one batched network evaluation per optimisation step ---------------------------------------------------------------
    # A dynamic mapping iteration (utils/slam_backend.py:336-771) asks the node network for 4-6 time samples per view (the view's
    # deltas, its flow partner's, the ARAP / elastic samples around it): ~60 evaluations of a 512-row MLP, ~40 launches each, plus
    # their backward. The nodes and weights only change at optimizer.step(), so all samples of an iteration are ONE batch.
    def begin_iteration(self, times, positions_only=(), blend=None):
        """`times`: samples whose every head is needed (a view's deltas, its flow partner's); `positions_only`: samples of which only the
        node positions are read (the ARAP / elastic regularisers) -- the trunk runs on all of them, the translation head too, the rotation /
        scaling / local-frame heads on the first group only. `blend` = (x, motion_mask): also warp these Gaussians at every sample of the
        first group, all in ONE blend launch each way (control_nodes.node_blend_batch) instead of one per view and flow partner; forward()
        then serves those samples from the batch."""
        full = sorted({time_key(t) for t in times})
        rest = sorted({time_key(t) for t in positions_only} - set(full))
        keys = full + rest
        if not keys or self.node_num == 0:
            self._batch = None
            return
        M, net = self.node_num, self.network
        if self.nodes.is_cuda and self.nodes.dtype == torch.float32 and os.environ.get("GSR_BATCH_TRUNK", "1") != "0":
            # On the device the batch goes through begin_iteration_indexed: the whole network as ONE autograd node on the dense kernels
            # (_FusedTrunk: embedding in one launch, the layers as one chain, the nine weight gradients as one launch), the heads as one
            # matrix, the blend packed -- the path the captured iterations take; only the bookkeeping by host time is this function's.
            # (The layer-by-layer form below ran the eager map() iterations and colour refinement on the library's GEMMs: 81 samples x 512
            # nodes per refinement iteration, a weight gradient of [256, 41 472] x [41 472, 84] at 146 us on a 32 x 64 macro tile.)
            it = self.begin_iteration_indexed(self._upload(keys), len(full), blend=blend)
            batch = {key: {} for key in keys}
            for group, rows in ((full, it["d_xyz_full"]), (rest, it["d_xyz_rest"])):
                if rows is not None:
                    for i, row in enumerate(rows.unbind(0)):
                        batch[group[i]]["d_xyz"] = row
            for name, t in it["heads"].items():
                for i, row in enumerate(t.unbind(0)):
                    batch[full[i]][name] = row
            self._batch = batch
            if it["blended"] is not None:
                rows = it["blended"]
                self._blended = {"n": int(blend[0].shape[0]), "masked": blend[1] is not None,
                                 "rows": {key: {"d_xyz": rows[0][i], "d_rotation": rows[1][i], "d_scaling": rows[2][i], "d_opacity": None, "d_color": None}
                                          for i, key in enumerate(full)}}
            self._graph = None
            return
        tt = self._upload(keys)[:, None]
        xe = _embed(self.nodes.detach(), net.multires)
        te = _embed(tt, net.t_multires)
        emb = torch.cat([xe[None].expand(len(keys), M, -1), te[:, None].expand(len(keys), M, -1)], -1)
        h = net.trunk(emb.reshape(len(keys) * M, -1))
        # unbind, not out[i]: one backward node per head (a stack of the slices' gradients) instead of one per slice, each of which would
        # zero-fill and accumulate a buffer of the whole batch (~250 launches per iteration at 90 time samples)
        self._batch = {key: {} for key in keys}
        self._blended = None
        d_xyz_all = net.gaussian_warp(h).reshape(len(keys), M, 3)
        for i, row in enumerate(d_xyz_all.unbind(0)):
            self._batch[keys[i]]["d_xyz"] = row
        if full:
            hf = h[:len(full) * M]
            heads = {"d_rotation": net.gaussian_rotation, "d_scaling": net.gaussian_scaling}
            if net.local_frame:
                heads["local_rotation"] = net.local_rotation
            stacked = {name: head(hf).reshape(len(full), M, -1) for name, head in heads.items()}
            for name, t in stacked.items():
                for i, row in enumerate(t.unbind(0)):
                    self._batch[full[i]][name] = row
            if blend is not None and blend[0] is not None and blend[0].shape[0] > 0:
                x, motion_mask = blend
                out = control_nodes.node_blend_batch(x, motion_mask, self.nodes, self._node_radius, self._node_weight, d_xyz_all[:len(full)],
                                                     stacked["d_rotation"], stacked["d_scaling"], stacked.get("local_rotation") if self.local_frame else None,
                                                     K=min(self.K, self.node_num), d_rot_as_res=self.d_rot_as_res, raw=True)
                rows = [t.unbind(0) for t in out]
                self._blended = {"n": int(x.shape[0]), "masked": motion_mask is not None,
                                 "rows": {key: {"d_xyz": rows[0][i], "d_rotation": rows[1][i], "d_scaling": rows[2][i], "d_opacity": None, "d_color": None}
                                          for i, key in enumerate(full)}}
        self._graph = None

    def begin_iteration_indexed(self, tt, n_full, blend=None):
        """begin_iteration for samples named by POSITION instead of by host value: `tt` is a device vector of times (float32 [n]); the first
        `n_full` need every head (and, with `blend` = (x, motion_mask), the Gaussians' deltas), the others only the node positions. The
        layout is the caller's and fixed -- no sorting, no merging of equal values -- which is what lets the times come from device
        memory the host never reads (slam/dynamic_graph.py: the random keyframes' times, the regularisers' random samples of a captured
        iteration). Returns {"d_xyz_full" [n_full, M, 3], "d_xyz_rest" [n - n_full, M, 3] (None when empty), "n_full", "n", "heads" {name:
        [n_full, M, .]}, "blended" (rows of d_xyz, d_rotation, d_scaling per full sample) or None}."""
        M, net = self.node_num, self.network
        n = int(tt.shape[0])
        if self.nodes.is_cuda and self.nodes.dtype == torch.float32 and os.environ.get("GSR_FUSED_EMBEDDING", "1") != "0":
            emb = control_nodes.node_embedding(self.nodes.detach(), tt, net.multires, net.t_multires)      # one launch instead of ~13
        else:
            xe = _embed(self.nodes.detach(), net.multires)
            te = _embed(tt.reshape(n, 1), net.t_multires)
            emb = torch.cat([xe[None].expand(n, M, -1), te[:, None].expand(n, M, -1)], -1)
        # (the heads as ONE linear layer on all rows: the rotation / scaling / local-frame columns of the position-only rows are computed and
        # never read -- 11 of 14 columns of a [rows, 256] x [256, 14] product)
        heads = net.heads()
        n_full = int(n_full)
        out = net.heads_from_embedding(emb.reshape(n * M, -1)).reshape(n, M, -1)
        # rows first (full samples | position-only samples), then the full samples' columns: the backward pass is two concatenations and ONE
        # zero-filled buffer (the position-only rows' unused columns) -- slicing every head out of the whole matrix cost a zero-fill and a copy
        # per head plus the additions of the pieces
        full, rest = out.split([n_full, n - n_full], 0) if 0 < n_full < n else ((out, None) if n_full else (None, out))
        widths = [m.weight.shape[0] for _, m in heads]
        cols = full.split(widths, -1) if full is not None else None
        it = {"d_xyz_full": cols[0] if cols is not None else None, "d_xyz_rest": rest[..., :widths[0]] if rest is not None else None,
              "n_full": n_full, "n": n, "heads": {}, "blended": None, "blended_stacked": None}
        if n_full:
            stacked = it["heads"] = {name: c for (name, _), c in list(zip(heads, cols))[1:]}
            if blend is not None and blend[0] is not None and blend[0].shape[0] > 0:
                x, motion_mask = blend
                if self.local_frame and sum(widths) == 14 and os.environ.get("GSR_PACKED_BLEND", "1") != "0":
                    # the heads' [n_full, M, 14] matrix as it is (gsr_node_blend.attr_stride): no copy per head, one gradient matrix back
                    out = control_nodes.node_blend_batch_packed(x, motion_mask, self.nodes, self._node_radius, self._node_weight, full,
                                                                K=min(self.K, self.node_num), d_rot_as_res=self.d_rot_as_res, raw=True)
                else:
                    out = control_nodes.node_blend_batch(x, motion_mask, self.nodes, self._node_radius, self._node_weight, cols[0],
                                                         stacked["d_rotation"], stacked["d_scaling"], stacked.get("local_rotation") if self.local_frame else None,
                                                         K=min(self.K, self.node_num), d_rot_as_res=self.d_rot_as_res, raw=True)
                it["blended"] = [t.unbind(0) for t in out]
                it["blended_stacked"] = out                    # (d_xyz, d_rotation, d_scaling), each [n_full, n, .]: control_nodes.fan_out's input
        self._batch, self._blended, self._graph = {}, None, None       # ("inside an iteration": _elastic_neighbours keeps its graph until end_iteration)
        return it

    def regularisers_indexed(self, it, n_window, n_extra, weights, window_samples=(4, 8), extra_samples=(2, 8)):
        """The ARAP and elastic terms of a dynamic mapping iteration (utils/slam_backend.py:517-524,646-652) from an indexed batch whose
        position-only samples are laid out view by view, a view's ARAP samples in front of its elastic ones: n_window views with
        window_samples = (4, 8), then n_extra random keyframes with extra_samples = (2, 8). `weights` [n_window + n_extra]: 1e-3 / 1e-4.
        The same terms as arap_loss_batch / elastic_loss_batch on lists of host times, read as slices of the batch instead of stacks."""
        M = self.node_num
        base = self.nodes.detach()
        wa, we = window_samples
        ea, ee = extra_samples
        nw, nx = n_window * (wa + we), n_extra * (ea + ee)
        # (split, not slices: one backward node that concatenates the pieces' gradients instead of a zero-fill + copy per slice)
        n_rest = it["n"] - it["n_full"]
        pieces = it["d_xyz_rest"].split([nw, nx] + ([n_rest - nw - nx] if n_rest > nw + nx else []), 0)
        parts_e, reg = [], 0
        w_a = w_e = e_a = e_e = None
        if n_window:
            w_a, w_e = (base + pieces[0].reshape(n_window, wa + we, M, 3)).split([wa, we], 1)
            parts_e.append(w_e)
        if n_extra:
            e_a, e_e = (base + pieces[1].reshape(n_extra, ea + ee, M, 3)).split([ea, ee], 1)
            parts_e.append(e_e)
        nodes_t = (parts_e[0] if len(parts_e) == 1 else torch.cat(parts_e, 0)).permute(0, 2, 1, 3)             # [V, M, T, 3]
        nn_weight, nn_idx = self._elastic_neighbours()
        # (weighted sums as dot products: one launch each way instead of a product and a reduction; nobody reads the value's last bit)
        reg = torch.dot(elastic_error(nodes_t, nn_weight, nn_idx), weights)
        if n_window and n_extra and w_a.is_cuda:
            # the neighbour sets of both groups of views in one k-NN call and ONE set of reverse lists (the two groups differ in their number
            # of samples, not in how a view's neighbours are found: deform_utils.py:58-110 on each view's first sample)
            nn_i, keep = connectivity_from_points(torch.cat([w_a[:, 0].detach(), e_a[:, 0].detach()], 0), K=10)
            sets = control_nodes.IndexSets(nn_i.reshape(n_window + n_extra, -1), M)
            reg = reg + torch.dot(weights[:n_window], arap_error(w_a, nn_i[:n_window], keep[:n_window], sets=sets, set_offset=0))
            reg = reg + torch.dot(weights[n_window:], arap_error(e_a, nn_i[n_window:], keep[n_window:], sets=sets, set_offset=n_window))
            return reg
        if n_window:
            nn_i, keep = connectivity_from_points(w_a[:, 0], K=10)
            reg = reg + (weights[:n_window] * arap_error(w_a, nn_i, keep)).sum()
        if n_extra:
            nn_i, keep = connectivity_from_points(e_a[:, 0], K=10)
            reg = reg + (weights[n_window:] * arap_error(e_a, nn_i, keep)).sum()
        return reg

    def _upload(self, values):
        """A list of host floats as a device vector without blocking: through a small ring of pinned staging buffers (torch.tensor(...,
        device=) is a synchronous pageable copy: ~0.2 ms per call behind a busy queue, once per mapping iteration)."""
        if self.device.type != "cuda":
            return torch.tensor(values, dtype=torch.float32, device=self.device)
        ring = self.__dict__.setdefault("_staging", {"slots": [], "next": 0})
        if len(ring["slots"]) < 8:
            ring["slots"].append([torch.empty(1024, dtype=torch.float32).pin_memory(), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"] % 8]
            ring["next"] += 1
            slot[1].synchronize()                          # the copy that last used this buffer (eight uploads ago) has long finished
        n = len(values)
        if n > slot[0].numel():
            return torch.tensor(values, dtype=torch.float32, device=self.device)
        slot[0][:n] = torch.tensor(values, dtype=torch.float32)
        out = slot[0][:n].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def end_iteration(self):
        self._batch = None
        self._graph = None
        self._blended = None

    def node_deform(self, t, key=None):
        """:1038-1051: per-node translation / rotation / scale (/ local rotation) at time t [M,1] (key: the host-side value of t, see
        begin_iteration)."""
        o = self._batch.get(time_key(key)) if (key is not None and self._batch is not None) else None
        if o is None or "d_rotation" not in o:            # not part of the iteration's batch (or batched for its positions only)
            o = self.network(self.nodes.detach(), t)
        return o

    def forward(self, x, t, motion_mask=None, t_key=None, **_):
        """:1192-1258."""
        b = self._blended
        # (between begin_iteration and end_iteration neither the Gaussians nor the nodes move -- the caller's contract, as for the batched
        # network evaluation: the positions are the ones begin_iteration(blend=...) was given; only their count is checked)
        if b is not None and t_key is not None and b["n"] == int(x.shape[0]) and b["masked"] == (motion_mask is not None):
            hit = b["rows"].get(time_key(t_key))
            if hit is not None:                               # warped with the iteration's other samples in one launch (begin_iteration)
                return hit
        na = self.node_deform(t, t_key)
        out = control_nodes.node_blend(x, motion_mask, self.nodes, self._node_radius, self._node_weight, na["d_xyz"], na["d_rotation"],
                                       na["d_scaling"], na.get("local_rotation") if self.local_frame else None, K=min(self.K, self.node_num),
                                       d_rot_as_res=self.d_rot_as_res, raw=True)
        return {"d_xyz": out["d_xyz"], "d_rotation": out["d_rotation"], "d_scaling": out["d_scaling"], "d_opacity": None, "d_color": None}

    def _d_xyz_at(self, t):
        o = self._batch.get(time_key(t)) if self._batch is not None else None
        if o is None:
            o = self.node_deform(torch.full((self.node_num, 1), float(t), dtype=torch.float32, device=self.device))
        return o["d_xyz"]

    def node_positions(self, times):
        """nodes + d_xyz(t) for a list of host times -> [M, T, 3] (time_utils.py:1136-1137, 1153-1154); served from the iteration's batched
        evaluation when there is one."""
        return self.nodes.detach()[:, None, :] + torch.stack([self._d_xyz_at(t) for t in times], 1)

    def node_positions_many(self, times_per_view):
        """The same for several equally long lists of times in one stack + one add: [V, T, M, 3]."""
        V, T = len(times_per_view), len(times_per_view[0])
        flat = torch.stack([self._d_xyz_at(t) for ts in times_per_view for t in ts])
        return self.nodes.detach() + flat.reshape(V, T, self.node_num, 3)

    def arap_loss(self, times):
        """ControlNodeWarp.arap_loss (:1128-1141) on the time samples `times` (draw_loss_times): connectivity of the nodes at the first
        sample (K = 10), then the ARAP error of the other samples against it."""
        if self.node_num < 3:
            return torch.zeros((), device=self.device)
        return self.arap_loss_batch([times])[0]

    def arap_loss_batch(self, times_per_view):
        """arap_loss for several views with the same number of samples in one pass: [V] errors."""
        nodes_seq = self.node_positions_many(times_per_view)                                               # [V, T, M, 3]
        nn_idx, keep = connectivity_from_points(nodes_seq[:, 0], K=10)
        return arap_error(nodes_seq, nn_idx, keep)

    def _elastic_neighbours(self, K=2):
        if self._batch is not None and self._graph is not None:     # inside an iteration neither the nodes nor their radii move
            return self._graph
        kk = min(K + 1, self.node_num)
        w, _, idx = control_nodes.cal_nn_weight(self.nodes.detach(), self.nodes.detach(), self._node_radius, self._node_weight, K=kk)
        out = (w[:, 1:], idx[:, 1:])
        if self._batch is not None:
            self._graph = out
        return out

    def elastic_loss(self, times):
        """ControlNodeWarp.elastic_loss (:1143-1165, the paper's "APAR" term) on the time samples `times`."""
        if self.node_num < 3:
            return torch.zeros((), device=self.device)
        return self.elastic_loss_batch([times])[0]

    def elastic_loss_batch(self, times_per_view):
        nodes_t = self.node_positions_many(times_per_view).permute(0, 2, 1, 3)                             # [V, M, T, 3]
        nn_weight, nn_idx = self._elastic_neighbours()
        return elastic_error(nodes_t, nn_weight, nn_idx)


class DeformModel:
    """gaussian_splatting/scene/deform_model.py:20-118: holder of the node warp + its optimizer."""

    def __init__(self, K=3, node_num=512, d_rot_as_res=True, local_frame=True, lr=None, device="cuda", position_lr_init=0.00016, deform_lr_scale=1.0, **_):
        self.deform = ControlNodes(node_num=node_num, K=K, d_rot_as_res=d_rot_as_res, local_frame=local_frame, device=device)
        self.spatial_lr_scale = 5
        # train_setting (:34-42): Adam(eps 1e-15) at position_lr_init * 5 * deform_lr_scale for both groups; the SLAM loops never call
        # update_learning_rate, so the rate stays there (arguments.py:127,138)
        self.lr = position_lr_init * self.spatial_lr_scale * deform_lr_scale if lr is None else lr
        self.optimizer = None
        self.reg_loss = 0.0

    def train_setting(self, *_):
        groups = [{"params": g["params"], "lr": self.lr, "name": g["name"]} for g in self.deform.trainable_parameters()]
        # fused=True on the device: ONE multi-tensor kernel for the ~25 network tensors instead of torch's foreach chain (~14 launches,
        # 0.14 ms of device time per mapping iteration); the same Adam arithmetic
        on_device = all(p.is_cuda for g in groups for p in g["params"])
        # capturable: the step counters live on the device and the step can be recorded in a hipGraph (slam/dynamic_graph.py); same arithmetic
        if on_device and os.environ.get("GSR_NETWORK_ADAM", "1") != "0":
            # ... and its step as two launches of this library (fused_adam.DeviceCountAdam) instead of torch's two multi-tensor launches of 24 us
            from fused_adam import DeviceCountAdam
            self.optimizer = DeviceCountAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, **({"fused": True, "capturable": True} if on_device else {}))

    def step(self, x, time_input, iteration=0, feature=None, motion_mask=None, camera_center=None, time_interval=None, **kw):
        """deform_model.py:32-33."""
        return self.deform(x, time_input, motion_mask=motion_mask, t_key=kw.get("t_key"))

    def extend_node_from_point(self, init_pcl, **kw):
        """:71-95: first call initialises the nodes; later calls append the new rows to the three node tensors and extend their Adam
        moments with zeros."""
        d = self.deform
        if not d.inited:
            d.init(init_pcl)
            self.train_setting()
            return
        new = d.extend_node(init_pcl)
        if new is None:
            return
        names = ("nodes", "_node_radius", "_node_weight")
        group = next(g for g in self.optimizer.param_groups if g["name"] == "nodes")
        for i, (name, ext) in enumerate(zip(names, new)):
            old = group["params"][i]
            st = self.optimizer.state.pop(old, None)
            p = nn.Parameter(torch.cat((old.detach(), ext.to(old.dtype)), dim=0).requires_grad_(True))
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext, dtype=old.dtype)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext, dtype=old.dtype)), dim=0)
                self.optimizer.state[p] = st
            group["params"][i] = p
            setattr(d, name, p)
