"""fp32-accurate dense layers on the bf16 matrix cores (include/dense_layers.h) against fp64 products: the error of the six-term bf16 split must
be of the size of an fp32 GEMM's own rounding (the library's fp32 GEMM is measured beside it)."""
import numpy as np
import pytest
import torch

import dense_layers as dl

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-300))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (300, 256, 84), (20000, 256, 256), (777, 256, 340), (130, 14, 256), (129, 129, 33), (4096, 256, 256)])
@pytest.mark.parametrize("opts", [dict(), dict(bias=True, relu=True), dict(gate=True)])
def test_dense_forward_against_fp64(M, N, K, opts):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    X = torch.randn((M, K), generator=g).to(DEV) * torch.exp(torch.randn((M, 1), generator=g)).to(DEV)      # rows of very different magnitude
    W = (torch.randn((N, K), generator=g) / np.sqrt(K)).to(DEV)
    b = torch.randn((N,), generator=g).to(DEV) if opts.get("bias") else None
    gate = torch.randn((M, K), generator=g).to(DEV) if opts.get("gate") else None
    Y = dl.dense_forward(X, dl.split_weight(W), N, K, b, relu=bool(opts.get("relu")), gate=gate)
    Xe = X.double() * (gate > 0) if gate is not None else X.double()
    ref = Xe @ W.double().t() + (b.double() if b is not None else 0.0)
    ref = torch.relu(ref) if opts.get("relu") else ref
    lib32 = (Xe.float() @ W.t() + (b if b is not None else 0.0))
    lib32 = torch.relu(lib32) if opts.get("relu") else lib32
    err, err32 = rel(Y, ref), rel(lib32, ref)
    assert Y.shape == (M, N) and torch.isfinite(Y).all()
    assert err < 1e-6 and err < 4 * err32 + 2e-7, (err, err32)
    # element-wise: no entry off by more than a few fp32 roundings of the products' magnitude
    scale = Xe.abs() @ W.double().abs().t() + (b.double().abs() if b is not None else 0.0)
    assert float(((Y.double() - ref).abs() / scale.clamp_min(1e-30)).max()) < 2e-6


def test_dense_forward_strided_input_transposed_weight_and_asymmetry():
    """A column range of a wider matrix (the skip layer's [emb | h] input), the transposed planes (dX = G W), and an asymmetric product that a
    row <-> column swap of the result would not survive."""
    g = torch.Generator(device="cpu").manual_seed(3)
    wide = torch.randn((1000, 340), generator=g).to(DEV)
    W = torch.randn((256, 340), generator=g).to(DEV) / 18.0
    Y = dl.dense_forward(wide, dl.split_weight(W), 256, 340)
    assert rel(Y, wide.double() @ W.double().t()) < 1e-6
    # dX = G W[:, 84:]: planes of the transpose of a column range
    G = torch.randn((1000, 256), generator=g).to(DEV)
    dX = dl.dense_forward(G, dl.split_weight(W, k0=84, K=256, transposed=True), 256, 256)
    assert rel(dX, G.double() @ W.double()[:, 84:]) < 1e-6
    X = torch.eye(64, device=DEV)
    A = (torch.arange(64 * 48, device=DEV, dtype=torch.float32).view(48, 64) * 1e-3).contiguous()        # W [N = 48, K = 64], asymmetric
    out = dl.dense_forward(X, dl.split_weight(A), 48, 64)
    assert torch.allclose(out, A.t(), rtol=1e-6, atol=0)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (33, 256, 84), (20000, 256, 256), (5000, 256, 340), (2080, 14, 256), (100000, 256, 256)])
@pytest.mark.parametrize("gated", [False, True])
def test_dense_wgrad_against_fp64_and_reproducible(M, N, K, gated):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    G = torch.randn((M, N), generator=g).to(DEV)
    X = torch.randn((M, K), generator=g).to(DEV)
    gate = torch.randn((M, N), generator=g).to(DEV) if gated else None
    dW = dl.dense_wgrad(G, X, gate=gate)
    Ge = G.double() * (gate > 0) if gated else G.double()
    ref = Ge.t() @ X.double()
    err, err32 = rel(dW, ref), rel(Ge.float().t() @ X, ref)
    assert dW.shape == (N, K) and err < 2e-6 and err < 4 * err32 + 2e-7, (err, err32)
    assert torch.equal(dW, dl.dense_wgrad(G, X, gate=gate))                        # fixed summation order


@pytest.mark.parametrize("M", [1, 37, 2080, 33280, 100003])
def test_dense_wgrad_many_against_fp64_reproducible_and_transpose_detecting(M):
    """The node network's shapes in ONE launch (gsr_dense_wgrad_many): layer 0 [256, 84], a [256, 256] layer with strided operands (the skip
    buffer's column window), the skip layer [256, 340], the heads [14, 256] (rows not 16-byte aligned: element-wise loads) and an odd shape."""
    g = torch.Generator(device="cpu").manual_seed(M)
    shapes = [(256, 84), (256, 256), (256, 340), (14, 256), (129, 33)]
    pairs = []
    for n, k in shapes:
        G, X = torch.randn((M, n), generator=g).to(DEV), torch.randn((M, k), generator=g).to(DEV)
        pairs.append((G, X))
    wide = torch.randn((M, 340), generator=g).to(DEV)
    pairs[1] = (pairs[1][0], wide[:, 84:])                                        # row stride 340, first column 84: 16-byte aligned rows
    outs = dl.dense_wgrad_many(pairs)
    for (G, X), dW in zip(pairs, outs):
        ref = G.double().t() @ X.double()
        err, err32 = rel(dW, ref), rel(G.t() @ X, ref)
        assert dW.shape == ref.shape and err < 2e-6 and err < 4 * err32 + 2e-7, (tuple(dW.shape), err, err32)
    again = dl.dense_wgrad_many(pairs)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))                      # fixed summation order
    single = dl.dense_wgrad_many(pairs[2:3])[0]                                     # another slicing of the rows: same product to rounding
    assert rel(single, outs[2]) < 4e-6


def test_dense_wgrad_many_bad_arguments_raise():
    G, X = torch.randn((9, 4), device=DEV), torch.randn((10, 8), device=DEV)
    with pytest.raises(ValueError):
        dl.dense_wgrad_many([(G, X)])
    with pytest.raises(ValueError):
        dl.dense_wgrad_many([])
    with pytest.raises(ValueError):
        dl.dense_wgrad_many([(X, X)] * 13)


def test_dense_bad_arguments_raise():
    X = torch.randn((10, 8), device=DEV)
    W = torch.randn((4, 8), device=DEV)
    with pytest.raises(ValueError):
        dl.dense_forward(X[:, :4], dl.split_weight(W), 4, 8)
    with pytest.raises((ValueError, RuntimeError)):
        dl.dense_forward(X.cpu(), dl.split_weight(W), 4, 8)
    with pytest.raises(ValueError):
        dl.dense_wgrad(torch.randn((9, 4), device=DEV), X)


@pytest.mark.parametrize("R,E", [(1, 84), (64, 84), (1000, 84), (33280, 84), (1000, 64), (333, 32), (1000, 96), (257, 4)])
def test_layer_fused_trunk_forward_against_fp64(R, E):
    """gsr_trunk_forward (the node network's eight layers + heads in one launch, a 64-row tile's activations resident in LDS) against the
    same network in fp64: every layer's output and the heads; the library's fp32 layer-by-layer result is measured beside it. Embedding
    widths below 65 (multires = 8, t_multires = 6 gives 64) exercise the zero-padded embedding planes."""
    g = torch.Generator(device="cpu").manual_seed(R)
    W, D, skip = 256, 8, 4
    emb = torch.randn((R, E), generator=g).to(DEV)
    Ws, bs = [], []
    for k in range(D):
        K = E if k == 0 else (E + W if k == skip + 1 else W)
        Ws.append((torch.randn((W, K), generator=g) * np.sqrt(2.0 / K)).to(DEV))
        bs.append((torch.randn((W,), generator=g) * 0.1).to(DEV))
    Wh, bh = (torch.randn((14, W), generator=g) * 0.05).to(DEV), torch.randn((14,), generator=g).to(DEV)
    assert dl.trunk_supported(emb, Ws, skip, Wh)
    heads, inputs, outs = dl.trunk_forward(emb, Ws, bs, Wh, bh)
    h64, h32 = emb.double(), emb
    for k in range(D):
        h64 = torch.relu(h64 @ Ws[k].double().t() + bs[k].double())
        h32 = torch.relu(h32 @ Ws[k].t() + bs[k])
        assert outs[k].shape == (R, W) and rel(outs[k], h64) < 2e-6, (k, rel(outs[k], h64), rel(h32, h64))
        assert torch.equal(inputs[k][:, -W:] if k == skip + 1 else inputs[k], emb if k == 0 else outs[k - 1])
        if k == skip:
            h64, h32 = torch.cat([emb.double(), h64], -1), torch.cat([emb, h32], -1)
            assert torch.equal(inputs[k + 1][:, :E], emb)
    ref = h64 @ Wh.double().t() + bh.double()
    assert heads.shape == (R, 14) and rel(heads, ref) < 2e-6, (rel(heads, ref), rel(h32 @ Wh.t() + bh, ref))


def test_node_network_with_the_layer_fused_forward_matches_the_library_path(monkeypatch):
    """slam.deform_model.NodeNetwork (the shipped structure: D = 8, W = 256, skip behind layer 4) through _FusedTrunk with the opt-in
    layer-fused forward (GSR_LAYER_FUSED_TRUNK=1 -> gsr_trunk_forward) against the default library GEMMs: same heads, and -- through the
    unchanged backward pass on the saved activations -- the same parameter gradients."""
    from slam import deform_model as dm
    torch.manual_seed(4)
    net = dm.NodeNetwork().to(DEV)
    with torch.no_grad():                                                     # heads start (almost) at zero in the reference: give them size
        for _, m in net.heads():
            m.weight.normal_(0, 0.05)
            m.bias.normal_(0, 0.1)
    x, t = torch.rand((3000, 3), device=DEV) - 0.5, torch.rand((3000, 1), device=DEV)
    emb = torch.cat([dm._embed(x, net.multires), dm._embed(t, net.t_multires)], -1)
    cot = torch.randn((3000, 14), device=DEV)
    res = {}
    monkeypatch.setattr(dm, "DENSE_TRUNK", False)
    for fused in (False, True):
        monkeypatch.setattr(dm, "LAYER_FUSED_TRUNK", fused)
        for p in net.parameters():
            p.grad = None
        out = net.heads_from_embedding(emb)
        (out * cot).sum().backward()
        res[fused] = (out.detach(), {k: p.grad.clone() for k, p in net.named_parameters()})
    assert rel(res[True][0], res[False][0]) < 2e-6
    # gradients: both forwards are fp32-accurate, but a pre-activation within rounding of zero gets a different ReLU mask in the two -- a handful
    # of flipped entries among 3000 x 256 x 8, whose effect grows towards the first layer (measured 1.2e-3 there, 1e-6 at the heads)
    for k in res[False][1]:
        assert rel(res[True][1][k], res[False][1][k]) < 1e-2, (k, rel(res[True][1][k], res[False][1][k]))
    assert rel(res[True][1]["gaussian_warp.weight"], res[False][1]["gaussian_warp.weight"]) < 1e-5


@pytest.mark.parametrize("M", [1, 50, 777, 4176, 20000, 33280])
@pytest.mark.parametrize("masked", [True, False])
def test_dense_backward_input_mask_and_bias_gradient(M, masked):
    """gsr_dense_backward_input: dX = (G W) [mask > 0] and its column sums in one pass, at row counts that take different row tiles per block
    (dense_row_tiles: 2 .. 10), with the mask a column range of a wider matrix (the skip layer's output inside [emb | h])."""
    g = torch.Generator(device="cpu").manual_seed(M)
    N = K = 256
    G = torch.randn((M, K), generator=g).to(DEV)
    W = (torch.randn((K, N + 84), generator=g) / 16).to(DEV)                  # [out = K, in = 84 + N]: the skip layer's weight
    wide = torch.randn((M, N + 84), generator=g).to(DEV)
    mask = wide[:, 84:] if masked else None
    planes_t = dl.split_weight(W, k0=84, K=N, transposed=True)
    dX, db = dl.dense_backward_input(G, planes_t, N, K, mask=mask)
    ref = G.double() @ W.double()[:, 84:]
    if masked:
        ref = ref * (mask > 0)
    assert dX.shape == (M, N) and db.shape == (N,)
    assert rel(dX, ref) < 1e-6
    if masked:
        assert torch.equal(dX == 0, ~(mask > 0) | (dX == 0)) and bool((dX[~(mask > 0)] == 0).all())
    assert rel(db, ref.sum(0)) < 1e-5 and float((db.double() - dX.double().sum(0)).abs().max()) <= 1e-5 * float(dX.abs().sum(0).max()) + 1e-30
    dX2, db2 = dl.dense_backward_input(G, planes_t, N, K, mask=mask)
    assert torch.equal(dX, dX2) and torch.equal(db, db2)                         # fixed summation order
    dX3, none = dl.dense_backward_input(G, planes_t, N, K, mask=mask, want_bias=False)
    assert none is None and torch.equal(dX3, dX)


def test_split_weights_in_one_launch_equal_the_single_splits():
    g = torch.Generator(device="cpu").manual_seed(11)
    Ws = [torch.randn((256, k), generator=g).to(DEV) for k in (84, 256, 340, 256)] + [torch.randn((14, 256), generator=g).to(DEV)]
    requests = [(w, 0, None, False) for w in Ws] + [(Ws[2], 84, 256, True), (Ws[1], 0, 256, True), (Ws[4], 0, None, True)]
    views, buf = dl.split_weights(requests)
    for (w, k0, K, tr), v in zip(requests, views):
        single = dl.split_weight(w, k0=k0, K=K, transposed=tr)
        assert torch.equal(v[:single.numel()], single)
    views2, buf2 = dl.split_weights(requests, out=buf)
    assert buf2.data_ptr() == buf.data_ptr()
    with pytest.raises(RuntimeError):
        dl.split_weights([(Ws[0], 0, None, False)] * 25)


def test_dense_chain_equals_the_separate_launches():
    """gsr_dense_chain: eight products in one launch, each reading its predecessor's output (with bias + ReLU: the network's forward; with a
    mask and column sums: its input-gradient chain) -- bit-identical to one launch per product, at row counts with a ragged last block."""
    g = torch.Generator(device="cpu").manual_seed(21)
    for M in (1, 700, 33280):
        X0 = torch.randn((M, 84), generator=g).to(DEV)
        Ws = [(torch.randn((256, 84 if i == 0 else 256), generator=g) / 12).to(DEV) for i in range(8)]
        bs = [torch.randn((256,), generator=g).to(DEV) for _ in range(8)]
        planes = [dl.split_weight(w) for w in Ws]
        planes_t = [dl.split_weight(w, transposed=True) for w in Ws[1:]]
        ys = [torch.empty((M, 256), device=DEV) for _ in range(8)]
        dl.dense_chain([dict(X=X0 if i == 0 else ys[i - 1], planes=planes[i], K=int(Ws[i].shape[1]), bias=bs[i], relu=True, Y=ys[i]) for i in range(8)])
        h = X0
        for i in range(8):
            h = dl.dense_forward(h, planes[i], 256, int(Ws[i].shape[1]), bs[i], relu=True)
            assert torch.equal(ys[i], h), (M, i)
        # the way back: G_{i-1} = (G_i W_i) [y_{i-1} > 0], db_{i-1} = column sums
        G7 = torch.randn((M, 256), generator=g).to(DEV)
        Gs = [torch.empty((M, 256), device=DEV) for _ in range(7)]
        dbs = [torch.empty((256,), device=DEV) for _ in range(7)]
        ops, src = [], G7
        for i in range(7, 0, -1):
            ops.append(dict(X=src, planes=planes_t[i - 1], K=256, Y=Gs[i - 1], mask=ys[i - 1], dbias=dbs[i - 1]))
            src = Gs[i - 1]
        dl.dense_chain(ops)
        src = G7
        for i in range(7, 0, -1):
            want, wdb = dl.dense_backward_input(src, planes_t[i - 1], 256, 256, mask=ys[i - 1])
            assert torch.equal(Gs[i - 1], want) and torch.equal(dbs[i - 1], wdb), (M, i)
            src = want
    with pytest.raises(ValueError):
        dl.dense_chain([])


@pytest.mark.parametrize("chain", [True, False])
def test_node_network_on_the_dense_layers_matches_the_library_path(monkeypatch, chain):
    """The default trunk (slam.deform_model.DENSE_TRUNK: per-layer products on the bf16 matrix cores, mask + bias gradient in the
    input-gradient product's epilogue) against the library path (GSR_DENSE_TRUNK=0): heads to fp32-GEMM accuracy, parameter gradients up to the
    ReLU-mask flips of pre-activations within rounding of zero; and bit-reproducible from call to call."""
    from slam import deform_model as dm
    torch.manual_seed(5)
    net = dm.NodeNetwork().to(DEV)
    with torch.no_grad():
        for _, m in net.heads():
            m.weight.normal_(0, 0.05)
            m.bias.normal_(0, 0.1)
    R = 4176
    x, t = torch.rand((R, 3), device=DEV) - 0.5, torch.rand((R, 1), device=DEV)
    emb = torch.cat([dm._embed(x, net.multires), dm._embed(t, net.t_multires)], -1)
    cot = torch.randn((R, 14), device=DEV)
    monkeypatch.setattr(dm, "LAYER_FUSED_TRUNK", False)

    monkeypatch.setattr(dm, "DENSE_CHAIN", chain)

    def run(dense):
        monkeypatch.setattr(dm, "DENSE_TRUNK", dense)
        for p in net.parameters():
            p.grad = None
        out = net.heads_from_embedding(emb)
        (out * cot).sum().backward()
        return out.detach(), {k: p.grad.clone() for k, p in net.named_parameters()}

    lib, dense, again = run(False), run(True), run(True)
    assert rel(dense[0], lib[0]) < 2e-6
    for k in lib[1]:
        assert rel(dense[1][k], lib[1][k]) < 1e-2, (k, rel(dense[1][k], lib[1][k]))
        assert torch.equal(dense[1][k], again[1][k]), k
    assert rel(dense[1]["gaussian_warp.weight"], lib[1]["gaussian_warp.weight"]) < 1e-5
    assert torch.equal(dense[0], again[0])
    # against fp64 autograd of the same network: the dense path is no farther from it than the library path (both see the same few mask flips)
    net64 = dm.NodeNetwork().to(DEV).double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    h = emb.double()
    for i, layer in enumerate(net64.linear):
        h = torch.relu(layer(h))
        if i in net64.skips:
            h = torch.cat([emb.double(), h], -1)
    out64 = torch.cat([m(h) for _, m in net64.heads()], -1)
    (out64 * cot.double()).sum().backward()
    g64 = {k: p.grad for k, p in net64.named_parameters()}
    assert rel(dense[0], out64) < 2e-6
    for k in lib[1]:
        assert rel(dense[1][k], g64[k]) < max(4 * rel(lib[1][k], g64[k]), 1e-5), (k, rel(dense[1][k], g64[k]), rel(lib[1][k], g64[k]))
