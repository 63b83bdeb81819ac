"""fp32-accurate dense layers on the bf16 matrix cores (include/dense_layers.h, csrc/gs_dense.h): host side.

The node network's trunk (utils/time_utils.py:327-476: eight layers of width 256 over every (node, time sample) row of a mapping iteration) is
GEMM-shaped work that the fp32 matrix instructions run at the fp32 VECTOR rate. Here an fp32 operand travels as three bf16 terms (24 bits of
significand, split by truncation) and a product as its six largest cross terms on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the results
are fp32 GEMM results (what is dropped is below 2e-7 of |x w| per product), at 1.4x (33 k rows) to 1.9x (66 k) the library's fp32 GEMM rate.
slam/deform_model._FusedTrunk runs the trunk's forward and input-gradient products through this module.

    planes = split_weight(W)                          # once per optimizer step: bf16 planes of W [N, K] ...
    planes_t = split_weight(W, transposed=True)       # ... and of its transpose (for the input gradient)
    Y = dense_forward(X, planes, N, K, bias, relu=True)
    dX = dense_forward(G, planes_t, K, N)             # G W
    dX, db = dense_backward_input(G, planes_t, K, N, mask=Y_below)   # (G W) [Y_below > 0] and its column sums: the layer below's G and bias gradient
    dW = dense_wgrad(G, X)                            # G^T X, deterministic

There is no CPU path: tensors must be fp32 on a HIP device."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C

_declared = False


class _Trunk(C.Structure):          # gsr_trunk
    _fields_ = [("E", C.c_int32), ("n_head_outputs", C.c_int32), ("planes", C.c_void_p * 10), ("bias", C.c_void_p * 9)]


class _ChainOp(C.Structure):         # gsr_dense_chain_op
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int32), ("K", C.c_int32), ("planes", C.c_void_p), ("bias", C.c_void_p), ("relu", C.c_int32),
                ("Y", C.c_void_p), ("ldy", C.c_int32), ("mask", C.c_void_p), ("ldmask", C.c_int32), ("dbias", C.c_void_p)]


class _WgradItem(C.Structure):      # gsr_dense_wgrad_item
    _fields_ = [("G", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("ldg", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32),
                ("N", C.c_int32), ("K", C.c_int32)]


class _SplitItem(C.Structure):      # gsr_dense_split_item
    _fields_ = [("W", C.c_void_p), ("planes", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("ldw", C.c_int32), ("k0", C.c_int32), ("transposed", C.c_int32)]


def _lib():
    global _declared
    lib = _C.load_library()
    if not _declared:
        i, vp = C.c_int, C.c_void_p
        lib.gsr_dense_planes_size.restype = C.c_size_t
        lib.gsr_dense_planes_size.argtypes = [i, i]
        lib.gsr_dense_split.restype = i
        lib.gsr_dense_split.argtypes = [i, i, vp, i, i, i, vp, vp]
        lib.gsr_dense_forward.restype = i
        lib.gsr_dense_forward.argtypes = [i, i, i, vp, i, vp, i, vp, vp, i, vp, i, vp]
        lib.gsr_dense_wgrad_workspace_size.restype = C.c_size_t
        lib.gsr_dense_wgrad_workspace_size.argtypes = [i, i, i]
        lib.gsr_dense_wgrad.restype = i
        lib.gsr_dense_wgrad.argtypes = [i, i, i, vp, i, vp, i, vp, i, vp, i, vp, vp]
        lib.gsr_dense_wgrad_many_workspace_size.restype = C.c_size_t
        lib.gsr_dense_wgrad_many_workspace_size.argtypes = [i, i, C.POINTER(_WgradItem)]
        lib.gsr_dense_wgrad_many.restype = i
        lib.gsr_dense_wgrad_many.argtypes = [i, i, C.POINTER(_WgradItem), vp, vp]
        lib.gsr_dense_backward_input_workspace_size.restype = C.c_size_t
        lib.gsr_dense_backward_input_workspace_size.argtypes = [i, i]
        lib.gsr_dense_backward_input.restype = i
        lib.gsr_dense_backward_input.argtypes = [i, i, i, vp, i, vp, vp, i, vp, i, vp, vp, vp]
        lib.gsr_dense_chain_workspace_size.restype = C.c_size_t
        lib.gsr_dense_chain_workspace_size.argtypes = [i, i, i]
        lib.gsr_dense_chain.restype = i
        lib.gsr_dense_chain.argtypes = [i, i, i, C.POINTER(_ChainOp), vp, vp]
        lib.gsr_dense_split_many.restype = i
        lib.gsr_dense_split_many.argtypes = [i, C.POINTER(_SplitItem), vp]
        lib.gsr_trunk_forward.restype = i
        lib.gsr_trunk_forward.argtypes = [C.POINTER(_Trunk), i, vp, C.POINTER(C.c_void_p), C.POINTER(i), vp, vp]
        _declared = True
    return lib


def _rows(t, name):
    _C._require_device(t, name)
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: fp32 [rows, cols] with unit column stride expected, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t


def split_weight(W, k0=0, K=None, transposed=False, out=None):
    """bf16 planes (a uint8 tensor) of W[:, k0:k0 + K] ([N, K], nn.Linear layout) for dense_forward -- or of its transpose."""
    W = _rows(W, "weight")
    N = int(W.shape[0])
    K = int(W.shape[1]) - k0 if K is None else int(K)
    lib = _lib()
    rows, cols = (K, N) if transposed else (N, K)
    size = int(lib.gsr_dense_planes_size(rows, cols))
    if out is None or out.numel() != size:
        out = torch.empty((size,), dtype=torch.uint8, device=W.device)
    with torch.cuda.device(W.device):
        rc = lib.gsr_dense_split(N, K, W.data_ptr(), int(W.stride(0)), int(k0), 1 if transposed else 0, out.data_ptr(), _C._stream(W.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_split")
    return out


def dense_forward(X, planes, N, K, bias=None, relu=False, gate=None, out=None):
    """Y [M, N] = act(X[:, :K] Wt + bias) for the split weight `planes` (N outputs, K inputs); gate [M, K]: X is read as X * (gate > 0)."""
    X = _rows(X, "X")
    M = int(X.shape[0])
    if int(X.shape[1]) < K:
        raise ValueError(f"X has {X.shape[1]} columns, the weight {K} inputs")
    if gate is not None:
        gate = _rows(gate, "gate")
        if gate.shape[0] != M or gate.shape[1] < K:
            raise ValueError("gate must cover X")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=X.device)
    lib = _lib()
    with torch.cuda.device(X.device):
        rc = lib.gsr_dense_forward(M, int(N), int(K), X.data_ptr(), int(X.stride(0)) if M > 1 else max(int(X.shape[1]), K),
                                   None if gate is None else gate.data_ptr(), 0 if gate is None else (int(gate.stride(0)) if M > 1 else int(gate.shape[1])),
                                   planes.data_ptr(), None if bias is None else bias.data_ptr(), 1 if relu else 0, out.data_ptr(),
                                   int(out.stride(0)) if M > 1 else int(out.shape[1]), _C._stream(X.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_forward")
    return out


def planes_size(W, k0=0, K=None, transposed=False):
    N = int(W.shape[0])
    K = int(W.shape[1]) - k0 if K is None else int(K)
    return int(_lib().gsr_dense_planes_size(*((K, N) if transposed else (N, K))))


def split_weights(requests, out=None):
    """Several split_weight calls as ONE launch (gsr_dense_split_many): requests = [(W, k0, K or None, transposed)], at most 24. Returns the
    planes of each request (views of one uint8 buffer, 256-byte aligned; `out`: a buffer of a previous call with the same requests, reused)."""
    sizes = [(planes_size(W, k0, K, tr) + 255) // 256 * 256 for (W, k0, K, tr) in requests]
    dev = requests[0][0].device
    if out is None or out.numel() != sum(sizes):
        out = torch.empty((sum(sizes),), dtype=torch.uint8, device=dev)
    items = (_SplitItem * len(requests))()
    views, off = [], 0
    for j, ((W, k0, K, tr), size) in enumerate(zip(requests, sizes)):
        W = _rows(W, "weight")
        views.append(out[off:off + size])
        it = items[j]
        it.W, it.planes, it.N = W.data_ptr(), views[-1].data_ptr(), int(W.shape[0])
        it.K, it.ldw, it.k0, it.transposed = (int(W.shape[1]) - k0 if K is None else int(K)), int(W.stride(0)), int(k0), 1 if tr else 0
        off += size
    lib = _lib()
    with torch.cuda.device(dev):
        rc = lib.gsr_dense_split_many(len(requests), items, _C._stream(dev))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_split_many")
    return views, out


def dense_backward_input(G, planes_t, N, K, mask=None, want_bias=True, out=None):
    """(dX, dbias) with dX [M, N] = (G [M, K] W) * (mask > 0) and dbias [N] its column sums (None without want_bias) -- the input gradient of a
    layer whose weight W is [K, N] (planes_t = its transposed planes), masked by the OUTPUT `mask` of the layer below and summed into that
    layer's bias gradient, in one pass (gsr_dense_backward_input; deterministic)."""
    G = _rows(G, "G")
    M = int(G.shape[0])
    if mask is not None:
        mask = _rows(mask, "mask")
        if int(mask.shape[0]) != M or int(mask.shape[1]) < N:
            raise ValueError("mask must cover dX")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=G.device)
    lib = _lib()
    dbias = torch.empty((N,), dtype=torch.float32, device=G.device) if want_bias else None
    ws = torch.empty((int(lib.gsr_dense_backward_input_workspace_size(M, N)),), dtype=torch.uint8, device=G.device) if want_bias else None
    ld = lambda t, cols: int(t.stride(0)) if M > 1 else int(cols)
    with torch.cuda.device(G.device):
        rc = lib.gsr_dense_backward_input(M, int(N), int(K), G.data_ptr(), ld(G, G.shape[1]), planes_t.data_ptr(), None if mask is None else mask.data_ptr(),
                                          0 if mask is None else ld(mask, mask.shape[1]), out.data_ptr(), ld(out, out.shape[1]),
                                          None if dbias is None else dbias.data_ptr(), None if ws is None else ws.data_ptr(), _C._stream(G.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_backward_input")
    return out, dbias


CHAIN_MAX, CHAIN_WIDTH = 8, 256


def dense_chain(ops):
    """Up to eight products of 256 output columns in a row on the same rows, in ONE launch (gsr_dense_chain): ops = [dict(X=, planes=, K=,
    bias=None, relu=False, Y=, mask=None, dbias=None)], op l + 1 usually reading op l's Y. X / Y / mask: fp32 [M, .] device tensors (column
    ranges of wider matrices allowed), dbias: fp32 [256] tensors that receive the column sums of Y. Returns nothing (the ops' Y / dbias are
    written)."""
    if not 1 <= len(ops) <= CHAIN_MAX:
        raise ValueError(f"dense_chain: 1..{CHAIN_MAX} products")
    M = int(ops[0]["X"].shape[0])
    dev = ops[0]["X"].device
    arr = (_ChainOp * len(ops))()
    ld = lambda t: int(t.stride(0)) if M > 1 else int(t.shape[1])
    want_ws = False
    for o, q in zip(arr, ops):
        X, Y = _rows(q["X"], "X"), _rows(q["Y"], "Y")
        if int(X.shape[0]) != M or int(Y.shape[0]) != M or int(Y.shape[1]) != CHAIN_WIDTH or int(X.shape[1]) < int(q["K"]):
            raise ValueError("dense_chain: every product maps [M, >= K] to [M, 256]")
        o.X, o.ldx, o.K, o.planes = X.data_ptr(), ld(X), int(q["K"]), q["planes"].data_ptr()
        bias, mask, dbias = q.get("bias"), q.get("mask"), q.get("dbias")
        o.bias, o.relu = (None if bias is None else bias.data_ptr()), (1 if q.get("relu") else 0)
        o.Y, o.ldy = Y.data_ptr(), ld(Y)
        if mask is not None:
            mask = _rows(mask, "mask")
        o.mask, o.ldmask = (None if mask is None else mask.data_ptr()), (0 if mask is None else ld(mask))
        o.dbias = None if dbias is None else dbias.data_ptr()
        want_ws = want_ws or dbias is not None
    lib = _lib()
    ws = torch.empty((int(lib.gsr_dense_chain_workspace_size(M, CHAIN_WIDTH, len(ops))),), dtype=torch.uint8, device=dev) if want_ws else None
    with torch.cuda.device(dev):
        rc = lib.gsr_dense_chain(M, CHAIN_WIDTH, len(ops), arr, None if ws is None else ws.data_ptr(), _C._stream(dev))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_chain")


def dense_wgrad(G, X, gate=None, out=None):
    """dW [N, K] = G^T X for G [M, N] (optionally gated: G * (gate > 0)) and X [M, K]; deterministic (row slices added in a fixed order)."""
    G, X = _rows(G, "G"), _rows(X, "X")
    M, N, K = int(G.shape[0]), int(G.shape[1]), int(X.shape[1])
    if int(X.shape[0]) != M:
        raise ValueError("G and X must have the same number of rows")
    if gate is not None:
        gate = _rows(gate, "gate")
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=G.device)
    lib = _lib()
    ws = torch.empty((int(lib.gsr_dense_wgrad_workspace_size(M, N, K)),), dtype=torch.uint8, device=G.device)
    ld = lambda t: int(t.stride(0)) if M > 1 else int(t.shape[1])
    with torch.cuda.device(G.device):
        rc = lib.gsr_dense_wgrad(M, N, K, G.data_ptr(), ld(G), None if gate is None else gate.data_ptr(), 0 if gate is None else ld(gate), X.data_ptr(), ld(X),
                                 out.data_ptr(), int(out.stride(0)), ws.data_ptr(), _C._stream(G.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_wgrad")
    return out


WGRAD_MANY_MAX = 12


def dense_wgrad_many(pairs, outs=None):
    """dW_i = G_i^T X_i for up to 12 (G_i [M, N_i], X_i [M, K_i]) pairs over the SAME M rows in ONE launch (gsr_dense_wgrad_many: every 128 x 128
    result tile x row slice resident at once, slices added in a fixed order). Returns the list of dW_i [N_i, K_i]."""
    if not 1 <= len(pairs) <= WGRAD_MANY_MAX:
        raise ValueError("dense_wgrad_many: 1..12 products")
    pairs = [(_rows(G, "G"), _rows(X, "X")) for G, X in pairs]
    M = int(pairs[0][0].shape[0])
    dev = pairs[0][0].device
    if any(int(G.shape[0]) != M or int(X.shape[0]) != M for G, X in pairs):
        raise ValueError("dense_wgrad_many: every G and X must have the same number of rows")
    if outs is None:
        outs = [torch.empty((int(G.shape[1]), int(X.shape[1])), dtype=torch.float32, device=dev) for G, X in pairs]
    ld = lambda t: int(t.stride(0)) if M > 1 else int(t.shape[1])
    arr = (_WgradItem * len(pairs))()
    for it, (G, X), o in zip(arr, pairs, outs):
        it.G, it.X, it.dW = G.data_ptr(), X.data_ptr(), o.data_ptr()
        it.ldg, it.ldx, it.lddw, it.N, it.K = ld(G), ld(X), int(o.stride(0)), int(G.shape[1]), int(X.shape[1])
    lib = _lib()
    ws = torch.empty((int(lib.gsr_dense_wgrad_many_workspace_size(M, len(pairs), arr)),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gsr_dense_wgrad_many(M, len(pairs), arr, ws.data_ptr(), _C._stream(dev))
    if rc < 0:
        _C._err(lib, rc, "gsr_dense_wgrad_many")
    return outs


TRUNK_LAYERS, TRUNK_WIDTH, TRUNK_SKIP, TRUNK_MAX_EMBEDDING, TRUNK_MAX_HEAD_OUTPUTS = 8, 256, 4, 96, 16


def trunk_supported(emb, weights, skip, W_heads):
    """Whether gsr_trunk_forward covers this network: the shipped structure (eight layers of width 256, the embedding re-injected behind layer
    4), an embedding of at most 96 columns whose width is a multiple of 4 (16-byte rows of the [emb | h] buffer), at most 16 head outputs."""
    E = int(emb.shape[1])
    return (len(weights) == TRUNK_LAYERS and skip == TRUNK_SKIP and E <= TRUNK_MAX_EMBEDDING and E % 4 == 0 and int(W_heads.shape[0]) <= TRUNK_MAX_HEAD_OUTPUTS
            and tuple(weights[0].shape) == (TRUNK_WIDTH, E) and tuple(weights[skip + 1].shape) == (TRUNK_WIDTH, E + TRUNK_WIDTH) and int(W_heads.shape[1]) == TRUNK_WIDTH
            and all(tuple(weights[k].shape) == (TRUNK_WIDTH, TRUNK_WIDTH) for k in range(1, TRUNK_LAYERS) if k != skip + 1))


def trunk_forward(emb, weights, biases, W_heads, b_heads):
    """The node network's trunk and heads in ONE launch (gsr_trunk_forward): returns (heads [R, n], inputs, outs) where outs[l] is layer l's
    output (post-ReLU) and inputs[l] the matrix layer l read -- emb, outs[l - 1], or for layer 5 the [R, E + 256] matrix [emb | outs[4]]
    (outs[4] is a column range of it) -- exactly what the layer-by-layer forward keeps for the backward pass. The weights are split into their
    bf16 planes on every call (they change with every optimizer step; ten small launches)."""
    emb = _rows(emb, "emb").contiguous()
    R, E = int(emb.shape[0]), int(emb.shape[1])
    dev = emb.device
    skip = TRUNK_SKIP
    # the kernel reads the two embedding products (layer 0, layer 5's first E columns) with a fixed K of TRUNK_MAX_EMBEDDING = 96 (three
    # 32-column steps, plane rows 96 apart); split_weight pads K to the next multiple of 32 only, so a narrower embedding (E <= 64) gets its
    # weight columns zero-padded to 96 here -- the kernel zero-fills the embedding's own columns [E, 96)
    def emb_planes(Wm):
        Wm = Wm[:, :E]
        if (E + 31) // 32 * 32 != TRUNK_MAX_EMBEDDING:
            Wm = torch.nn.functional.pad(Wm, (0, TRUNK_MAX_EMBEDDING - E))
        return split_weight(Wm, k0=0, K=int(Wm.shape[1]))
    planes = [emb_planes(weights[0])]
    planes += [split_weight(weights[k]) for k in range(1, skip + 1)]
    planes += [emb_planes(weights[skip + 1]), split_weight(weights[skip + 1], k0=E, K=TRUNK_WIDTH)]
    planes += [split_weight(weights[k]) for k in range(skip + 2, TRUNK_LAYERS)]
    planes.append(split_weight(W_heads))
    cat = torch.empty((R, E + TRUNK_WIDTH), dtype=torch.float32, device=dev)
    cat[:, :E] = emb
    outs = [cat[:, E:] if k == skip else torch.empty((R, TRUNK_WIDTH), dtype=torch.float32, device=dev) for k in range(TRUNK_LAYERS)]
    heads = torch.empty((R, int(W_heads.shape[0])), dtype=torch.float32, device=dev)
    t = _Trunk()
    t.E, t.n_head_outputs = E, int(W_heads.shape[0])
    bs = [b.contiguous() for b in biases] + [b_heads.contiguous()]
    for k, pl in enumerate(planes):
        t.planes[k] = pl.data_ptr()
    for k, b in enumerate(bs):
        t.bias[k] = b.data_ptr()
    out_ptrs = (C.c_void_p * TRUNK_LAYERS)(*[o.data_ptr() for o in outs])
    ldo = (C.c_int * TRUNK_LAYERS)(*[int(o.stride(0)) if R > 1 else int(E + TRUNK_WIDTH if k == skip else TRUNK_WIDTH) for k, o in enumerate(outs)])
    lib = _lib()
    with torch.cuda.device(dev):
        rc = lib.gsr_trunk_forward(C.byref(t), R, emb.data_ptr(), out_ptrs, ldo, heads.data_ptr(), _C._stream(dev))
    if rc < 0:
        _C._err(lib, rc, "gsr_trunk_forward")
    inputs = [emb] + [outs[k - 1] if k != skip + 1 else cat for k in range(1, TRUNK_LAYERS)]
    return heads, inputs, outs
