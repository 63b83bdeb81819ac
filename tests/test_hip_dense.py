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


def test_dense_bad_arguments_raise():
    X = torch.randn((10, 8), device=DEV)
    W = torch.randn((4, 8), device=DEV)
    with pytest.raises(ValueError):
        dl.dense_forward(X[:, :4], dl.split_weight(W), 4, 8)
    with pytest.raises((ValueError, RuntimeError)):
        dl.dense_forward(X.cpu(), dl.split_weight(W), 4, 8)
    with pytest.raises(ValueError):
        dl.dense_wgrad(torch.randn((9, 4), device=DEV), X)
