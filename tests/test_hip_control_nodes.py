"""GPU parity of the SC-GS control-node warp (include/control_nodes.h): gsr_knn_points, gsr_node_blend_forward / _backward against
the golden vectors of the reference's own ControlNodeWarp and against the pinned oracle (fp64) on seeded inputs.
Tolerances: values rel-L1 <= 2e-5, gradients rel-L1 <= 2e-4 (north_star: 1e-4 / 1e-3)."""
import os

import numpy as np
import pytest
import torch

import control_nodes as cn
from oracle import control_node_oracle as O
from test_control_node_oracle import G, load, check_against_golden, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_product(d, raw=True):
    if raw:                                                          # raw parameters: exp / sigmoid and their chain rules in the kernels
        radius, weight = d["node_radius_raw"], d["node_weight_raw"]
    else:                                                            # activated properties computed by torch (:893-898)
        radius, weight = torch.exp(d["node_radius_raw"]), torch.sigmoid(d["node_weight_raw"])
    res = cn.node_blend(d["x"], d["motion_mask"], d["nodes"], radius, weight, d["node_trans"], d["node_rot"], d["node_scale"],
                        d["local_rotation"] if d["local_frame"] else None, K=d["K"], d_rot_as_res=d["d_rot_as_res"], raw=raw)
    return res


@pytest.mark.parametrize("raw", [True, False])
@pytest.mark.parametrize("name", list(G["cases"]))
def test_node_blend_matches_reference_golden(name, raw):
    d = load(name, device=DEV)
    res = run_product(d, raw)
    check_against_golden(name, (res["d_xyz"], res["d_rotation"], res["d_scaling"]), d, (res["nn_weight"], res["nn_dist"], res["nn_idx"]))


@pytest.mark.parametrize("n,m,D,K", [(200, 200, 24, 9), (1000, 512, 3, 3), (333, 7, 3, 8), (50, 5000, 3, 32), (4097, 1300, 5, 4),
                                     (1, 1, 1, 1), (64, 40, 32, 16), (3000, 9000, 8, 5)])
def test_knn_points_against_oracle(n, m, D, K):
    rng = np.random.default_rng(n + m)
    p1 = torch.tensor(rng.normal(size=(1, n, D)).astype(np.float32), device=DEV)
    p2 = torch.tensor(rng.normal(size=(1, m, D)).astype(np.float32), device=DEV)
    out = cn.knn_points(p1, p2, None, None, K=K, return_nn=True)
    kk = min(K, m)
    dist, idx = O.knn_points(p1[0].double().cpu(), p2[0].double().cpu(), kk)
    got_d, got_i = out.dists[0, :, :kk].cpu(), out.idx[0, :, :kk].cpu()
    assert out.dists.shape == (1, n, K) and out.idx.dtype == torch.int64
    # the distances must match; the indices wherever fp32 could tell the candidates apart
    assert torch.allclose(got_d.double(), dist, rtol=1e-5, atol=1e-6)
    differ = got_i != idx
    if differ.any():
        d_at = ((p1[0].double().cpu()[:, None] - p2[0].double().cpu()[got_i]) ** 2).sum(-1)
        assert torch.allclose(d_at[differ], dist[differ], rtol=1e-5, atol=1e-6)
    assert float(differ.float().mean()) < 1e-3
    assert torch.all(got_d[:, 1:] >= got_d[:, :-1])                  # ascending
    if m < K:                                                        # pytorch3d's padding
        assert torch.all(out.dists[0, :, m:] == 0) and torch.all(out.idx[0, :, m:] == 0)
    assert torch.equal(out.knn[0, :, :kk], p2[0][out.idx[0, :, :kk]])


@pytest.mark.parametrize("B,n,m,D,K", [(7, 512, 512, 3, 11), (3, 700, 1024, 3, 16), (2, 100, 300, 4, 32), (4, 65, 5, 2, 9), (5, 300, 2000, 3, 11), (3, 400, 600, 3, 3)])
def test_knn_points_batched_equals_one_call_per_element(B, n, m, D, K):
    """pytorch3d's batched form (deform_utils.py:74) in one call; small candidate sets with K > 4 take the wave-per-query kernel, whose lists
    must be those of the thread-per-query kernel bit for bit (same distance arithmetic, ties to the lower index)."""
    import ctypes
    rng = np.random.default_rng(B * n + m)
    p1 = torch.tensor(rng.normal(size=(B, n, D)).astype(np.float32), device=DEV)
    p2 = torch.tensor(rng.normal(size=(B, m, D)).astype(np.float32), device=DEV)
    p2[:, 1::7] = p2[:, ::7][:, :p2[:, 1::7].shape[1]]               # exact duplicates among the candidates: ties
    out = cn.knn_points(p1, p2, K=K)
    lib = cn._lib()
    for b in range(B):
        d1 = torch.empty((n, K), dtype=torch.float32, device=DEV)
        i1 = torch.empty((n, K), dtype=torch.int64, device=DEV)
        assert lib.gsr_knn_points(n, m, D, K, p1[b].data_ptr(), p2[b].data_ptr(), d1.data_ptr(), i1.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(out.idx[b], i1) and torch.equal(out.dists[b], d1)
    kk = min(K, m)
    tie = out.dists[..., 1:kk] == out.dists[..., :kk - 1]            # among equal distances the lower index comes first (both kernels)
    assert bool(tie.any()) and bool((out.idx[..., 1:kk] > out.idx[..., :kk - 1])[tie].all())
    ref = torch.cdist(p1.double(), p2.double()) ** 2
    want = ref.topk(kk, dim=-1, largest=False).values
    assert torch.allclose(out.dists[..., :kk].double(), want, rtol=1e-5, atol=1e-6)
    if K <= 4:
        return
    nan = p1.clone()
    nan[0, 0, 0] = float("nan")                                      # a query with a NaN coordinate accepts nothing: pytorch3d's padding
    got = cn.knn_points(nan, p2, K=K)
    assert torch.all(got.idx[0, 0] == 0) and torch.all(got.dists[0, 0] == 0) and torch.equal(got.idx[0, 1:], out.idx[0, 1:])


def test_knn_points_golden_trajectories_and_self_match():
    p = torch.tensor(G["knn_traj/p"], device=DEV)
    out = cn.knn_points(p, p, None, None, K=9)
    assert np.array_equal(out.idx.cpu().numpy(), G["knn_traj/idx"])
    assert np.allclose(out.dists.cpu().numpy(), G["knn_traj/dist"], rtol=1e-5, atol=1e-6)
    dup = torch.zeros((1, 6, 3), device=DEV)                         # all distances equal: the lower index comes first
    assert torch.equal(cn.knn_points(dup, dup, K=4).idx[0], torch.arange(4, device=DEV).expand(6, 4))


@pytest.mark.parametrize("m,K,local_frame,rot_res,with_weight", [(900, 3, True, True, True), (5000, 5, False, False, True),
                                                                  (300, 8, True, False, False), (2, 3, True, True, True)])
def test_node_blend_against_oracle_fp64(m, K, local_frame, rot_res, with_weight):
    """sizes beyond the golden ones: the global-atomic backward path (m > 720), several node chunks (m > 4096), K up to 8,
    fewer nodes than K."""
    rng = np.random.default_rng(m + K)
    n = 4000
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=DEV, requires_grad=rg)
    x, mask = T(rng.uniform(-1, 1, size=(n, 3))), T(rng.uniform(0.2, 1, size=(n, 1)))
    nodes = T(rng.uniform(-1, 1, size=(m, 5)))                       # hyper coordinates present but unused (node_stride 5)
    leaves = dict(rr=T(np.log(rng.uniform(0.1, 0.5, size=m)), True), wr=T(rng.normal(size=(m, 1)), True),
                  tr=T(rng.normal(scale=0.1, size=(m, 3)), True), ro=T(rng.normal(scale=0.2, size=(m, 4)), True),
                  sc=T(rng.normal(scale=0.1, size=(m, 3)), True), lr=T(rng.normal(scale=0.3, size=(m, 4)), True))
    res = cn.node_blend(x, mask, nodes, leaves["rr"], leaves["wr"] if with_weight else None, leaves["tr"],
                        leaves["ro"], leaves["sc"], leaves["lr"] if local_frame else None, K=K, d_rot_as_res=rot_res)
    cots = [T(rng.normal(size=(n, c))) for c in (3, 4, 3)]
    (sum((res[k] * c).sum() for k, c in zip(("d_xyz", "d_rotation", "d_scaling"), cots))).backward()
    l64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in leaves.items()}
    if m < K:
        # fewer nodes than K: pytorch3d pads the neighbour list with (index 0, distance 0); restate that by appending copies of
        # node 0 placed ON each query point is not possible for all points at once, so check the padding and finiteness directly
        assert torch.all(res["nn_idx"][:, m:] == 0) and torch.all(res["nn_dist"][:, m:] == 0)
        assert torch.allclose(res["nn_weight"].sum(1), torch.ones(n, device=DEV), atol=1e-5)
        assert all(torch.isfinite(v.grad).all() for v in leaves.values() if v.grad is not None)
        return
    o = O.node_blend(x.double().cpu(), mask.double().cpu(), nodes.double().cpu(), l64["rr"], l64["wr"] if with_weight else None, l64["tr"],
                     l64["ro"], l64["sc"], l64["lr"], K, local_frame, rot_res)
    (sum((a * c.double().cpu()).sum() for a, c in zip(o, cots))).backward()
    for k, a in zip(("d_xyz", "d_rotation", "d_scaling"), o):
        assert rel(res[k].detach().cpu(), a.detach()) < 2e-5, k
    for k in leaves:
        if l64[k].grad is None:
            continue
        assert rel(leaves[k].grad.cpu(), l64[k].grad) < 2e-4, (k, rel(leaves[k].grad.cpu(), l64[k].grad))


def test_cal_nn_weight_alone_with_gradients():
    d = load("shipped", device=DEV)
    w, dist, idx = cn.cal_nn_weight(d["x"], d["nodes"], d["node_radius_raw"], d["node_weight_raw"], K=3)
    cot = torch.randn_like(w)
    (w * cot).sum().backward()
    x64 = d["x"].double().cpu()
    rr, wr = d["node_radius_raw"].detach().double().cpu().requires_grad_(True), d["node_weight_raw"].detach().double().cpu().requires_grad_(True)
    w64, dist64, idx64 = O.cal_nn_weight(x64, d["nodes"].double().cpu(), rr, wr, 3)
    (w64 * cot.double().cpu()).sum().backward()
    assert torch.equal(idx.cpu(), idx64) and rel(w.detach().cpu(), w64.detach()) < 2e-5
    assert rel(d["node_radius_raw"].grad.cpu(), rr.grad) < 2e-4 and rel(d["node_weight_raw"].grad.cpu(), wr.grad) < 2e-4


def test_edge_cases_and_errors():
    d = load("k1", device=DEV)
    empty = cn.node_blend(torch.zeros((0, 3), device=DEV), None, d["nodes"], d["node_radius_raw"], None, d["node_trans"],
                          d["node_rot"], d["node_scale"], None, K=1)
    assert empty["d_xyz"].shape == (0, 3) and empty["nn_idx"].shape == (0, 1)
    (empty["d_xyz"].sum() + empty["d_rotation"].sum()).backward()    # zero gradients, no fault
    assert float(d["node_trans"].grad.abs().sum()) == 0.0
    # deterministic backward (block partials summed in a fixed order)
    d = load("shipped", device=DEV)
    grads = []
    for _ in range(2):
        for v in d.values():
            if isinstance(v, torch.Tensor) and v.grad is not None:
                v.grad = None
        r = run_product(d)
        (r["d_xyz"].sum() + r["d_rotation"].square().sum() + r["d_scaling"].sum()).backward()
        grads.append([d[k].grad.clone() for k in ("node_trans", "node_rot", "node_scale", "local_rotation", "node_radius_raw")])
    # LDS float atomics inside a block are order-dependent; across blocks the sum is fixed: allow rounding-level differences only
    for a, b in zip(*grads):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError, match="K = 9"):
        cn.node_blend(d["x"], None, d["nodes"], d["node_radius_raw"], None, d["node_trans"], d["node_rot"], d["node_scale"], None, K=9)
    with pytest.raises(ValueError, match="node_rot must have shape"):
        cn.node_blend(d["x"], None, d["nodes"], d["node_radius_raw"], None, d["node_trans"], d["node_rot"][:, :3], d["node_scale"], None, K=3)
    with pytest.raises(NotImplementedError):
        cn.knn_points(d["x"][None], d["nodes"][None], torch.tensor([5]), None, K=3)
    with pytest.raises(Exception, match="HIP device|no CPU"):
        cn.knn_points(d["x"][None].cpu(), d["nodes"][None].cpu(), K=3)


@pytest.mark.parametrize("B,n,m,local_frame,with_weight,with_mask", [(12, 2500, 512, True, True, True), (3, 700, 900, False, False, False), (1, 300, 64, True, True, False)])
def test_node_blend_batch_equals_one_blend_per_sample(B, n, m, local_frame, with_weight, with_mask):
    """gsr_node_blend_forward_batch / _backward_batch: B sets of node attributes (the time samples of one mapping iteration) blended onto
    the same Gaussians in one launch per stage. Outputs are those of B node_blend calls bit for bit, the per-sample node
    gradients to rounding (the blocks accumulate with LDS float atomics), the radius / weight gradients are the sum of the B calls'."""
    g = torch.Generator(device="cpu").manual_seed(B * n + m)
    R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    x, nodes = R(n, 3), R(m, 3)
    radius = (R(m, sc=0.2) - 1.0).requires_grad_(True)
    weight = R(m, 1, sc=0.5).requires_grad_(True) if with_weight else None
    mask = (torch.rand(n, 1, generator=g) > 0.2).float().to(DEV) if with_mask else None
    attrs = [R(B, m, 3, sc=0.05).requires_grad_(True), R(B, m, 4, sc=0.05).requires_grad_(True), R(B, m, 3, sc=0.01).requires_grad_(True)]
    local = R(B, m, 4, sc=0.1).requires_grad_(True) if local_frame else None
    cots = [R(B, n, 3), R(B, n, 4), R(B, n, 3)]
    leaves = [radius] + ([weight] if with_weight else []) + attrs + ([local] if local_frame else [])

    out = cn.node_blend_batch(x, mask, nodes, radius, weight, attrs[0], attrs[1], attrs[2], local, K=3, d_rot_as_res=True, raw=True)
    torch.autograd.backward(list(out), cots)
    got = [t.detach().clone() for t in out] + [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    singles = []
    for b in range(B):
        r = cn.node_blend(x, mask, nodes, radius, weight, attrs[0][b], attrs[1][b], attrs[2][b], None if local is None else local[b], K=3, d_rot_as_res=True, raw=True)
        torch.autograd.backward([r["d_xyz"], r["d_rotation"], r["d_scaling"]], [c[b] for c in cots])
        singles.append(r)
    for k, name in enumerate(("d_xyz", "d_rotation", "d_scaling")):
        assert torch.equal(got[k], torch.stack([s_[name] for s_ in singles]))
    want = [t.grad for t in leaves]
    n_shared = 1 + int(with_weight)
    for a, b_ in zip(got[3:3 + n_shared], want[:n_shared]):          # radius / weight: summed over the samples (another association)
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-7)
    for a, b_ in zip(got[3 + n_shared:], want[n_shared:]):           # per-sample node gradients (LDS float atomics inside a block: order-dependent rounding)
        assert a.shape == b_.shape and torch.allclose(a, b_, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("rows,cols", [(50_000, 256), (77, 64), (1, 1024), (0, 256)])
def test_relu_backward_bias_matches_torch_and_is_reproducible(rows, cols):
    """gsr_relu_backward_bias: G = dY [Y > 0] exactly, the bias gradient = G's column sums (another association than torch's reduction:
    compared in fp64), the same bits on every call."""
    g = torch.Generator(device="cpu").manual_seed(rows + cols)
    dY = torch.randn(rows, cols, generator=g).to(DEV)
    Y = torch.relu(torch.randn(rows, cols, generator=g)).to(DEV)
    G, db = cn.relu_backward_bias(dY, Y)
    assert torch.equal(G, dY * (Y > 0))
    want = (dY.double() * (Y > 0)).sum(0)
    assert torch.allclose(db.double(), want, rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    G2, db2 = cn.relu_backward_bias(dY, Y)
    assert torch.equal(db, db2) and torch.equal(G, G2)
    with pytest.raises(ValueError):
        cn.relu_backward_bias(dY[:, :-1] if cols > 1 else dY, Y[:, :-1] if cols > 1 else Y[:0])


def test_fused_network_equals_the_op_by_op_network():
    """NodeNetwork.heads_from_embedding on the device (_FusedTrunk: the layers on the bf16 matrix cores with three-term operands, ReLU mask +
    bias gradient in the input-gradient product's epilogue, weight gradients batched over row groups, all heads as one layer) against the same
    network evaluated op by op (what runs on CPU tensors): values equal to GEMM rounding. Gradients: a pre-activation within rounding of zero
    may land on the other side of the ReLU in two evaluations -- a few dozen of the 68 M do, and each moves a gradient tensor by up to 1e-3 of
    its norm (the op-by-op fp32 network is that far from fp64 autograd itself). So the gradients are compared with fp64 autograd of the network
    evaluated WITH THE FUSED FORWARD'S OWN ReLU MASKS (its saved layer outputs): every gradient to 2e-5 of its norm; and with the op-by-op
    network as a whole to 5e-3."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "4dgs-slam_amd"))
    from slam.deform_model import NodeNetwork
    torch.manual_seed(3)
    net = NodeNetwork().to(DEV)
    for _, head in net.heads():                                  # (the shipped initialisation makes the heads ~0: give them something to propagate)
        torch.nn.init.normal_(head.weight, std=0.05)
    layers = list(net.linear) + [m for _, m in net.heads()]
    params = [p for layer in layers for p in (layer.weight, layer.bias)]
    D = len(net.linear)
    for rows in (6000, 33280, 6007):                             # (6007 is prime: the weight gradients fall back to single GEMMs)
        emb, cot = torch.randn(rows, net.input_ch, device=DEV), torch.randn(rows, 14, device=DEV)
        net.zero_grad(set_to_none=True)
        out = net.heads_from_embedding(emb)
        assert type(out.grad_fn).__name__.startswith("_FusedTrunk") and out.shape == (rows, 14)
        saved = out.grad_fn.saved_tensors                       # (W_heads, D weights, D layer inputs, D layer outputs)
        masks = [(t > 0).double() for t in saved[1 + 2 * D:1 + 3 * D]]
        out.backward(cot)
        got = [p.grad.clone() for p in params]
        # fp64 autograd of the same network with those masks
        p64 = [p.detach().double().requires_grad_(True) for p in params]
        h = emb.double()
        for i in range(D):
            h = (h @ p64[2 * i].t() + p64[2 * i + 1]) * masks[i]
            if i in net.skips:
                h = torch.cat([emb.double(), h], -1)
        out64 = torch.cat([h @ p64[2 * (D + k)].t() + p64[2 * (D + k) + 1] for k in range(len(net.heads()))], -1)
        assert float((out.double() - out64).norm()) <= 2e-6 * float(out64.norm())
        g64 = torch.autograd.grad(out64, p64, cot.double())
        for a, b in zip(got, g64):
            assert a.shape == b.shape and float((a.double() - b).norm()) <= 2e-5 * float(b.norm()), (rows, tuple(a.shape), float((a.double() - b).norm() / b.norm()))
        net.zero_grad(set_to_none=True)
        hh = net.trunk(emb)
        want_out = torch.cat([m(hh) for _, m in net.heads()], -1)
        assert torch.allclose(out, want_out, rtol=1e-4, atol=1e-4)
        want_out.backward(cot)
        for a, p in zip(got, params):
            b = p.grad
            assert float((a - b).norm()) <= 5e-3 * float(b.norm()), (rows, tuple(a.shape), float((a - b).norm() / b.norm()))


def test_node_embedding_equals_the_tensor_program():
    """gsr_node_embedding: the node network's [n * M, 84] input in one launch against the reference's embedders as tensor ops
    (slam.deform_model._embed, which the CPU tests compare with the reference's DeformNetwork): equal to 1 ulp of sin / cos."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "4dgs-slam_amd"))
    from slam.deform_model import _embed
    g = torch.Generator().manual_seed(2)
    nodes, tt = (torch.randn(137, 3, generator=g) * 0.7).to(DEV), torch.rand(9, generator=g).to(DEV)
    got = cn.node_embedding(nodes, tt, 10, 10)
    xe, te = _embed(nodes, 10), _embed(tt.reshape(9, 1), 10)
    want = torch.cat([xe[None].expand(9, 137, -1), te[:, None].expand(9, 137, -1)], -1).reshape(9 * 137, -1)
    assert got.shape == want.shape == (9 * 137, 84)
    assert torch.equal(got[:, :3], want[:, :3]) and torch.equal(got[:, 63], want[:, 63])
    assert float((got - want).abs().max()) <= 2e-6          # |sin|, |cos| <= 1: an ulp or two of the library functions at arguments up to 512 x
    assert cn.node_embedding(nodes, tt[:0], 10, 10).shape == (0, 84)


@pytest.mark.parametrize("S,E,Nv,C,B", [(1, 5120, 512, 6, 4), (3, 5120, 512, 3, 12), (7, 640, 64, 3, 7), (1, 30000, 512, 21, 2), (2, 16384, 1024, 3, 2),
                                        (1, 100, 1500, 3, 1), (1, 1, 1, 1, 1)])
def test_gather_rows_backward_is_an_ordered_scatter_add(S, E, Nv, C, B):
    """control_nodes.gather_rows: torch.gather's values and a backward pass that adds a target's incoming rows in a fixed order (gsr_index_csr:
    one block per set for small sets -- a stable counting sort --, one wave per target for large ones; gsr_segment_sum). Against an fp64
    index_add, bit-identical from call to call, targets that nobody points at included (Nv > distinct indices)."""
    import control_nodes as cn
    g = torch.Generator(device="cpu").manual_seed(S * 1000 + E)
    idx = torch.randint(0, max(1, Nv - 3), (S, E), generator=g).to(DEV)          # (the last targets receive nothing)
    table = torch.randn((B, Nv, C), generator=g).to(DEV).requires_grad_(True)
    set_of_b = None
    if S > 1 and S != B:
        set_of_b = torch.arange(S, device=DEV, dtype=torch.int32).repeat_interleave(B // S)
    sets = cn.IndexSets(idx, Nv)
    out = cn.gather_rows(table, sets, set_of_b)
    sel = idx if S == B else (idx.index_select(0, set_of_b.long()) if set_of_b is not None else idx.expand(B, -1))
    assert torch.equal(out, torch.gather(table.detach(), 1, sel[:, :, None].expand(-1, -1, C)))
    cot = torch.randn(out.shape, generator=g).to(DEV)
    (grad,) = torch.autograd.grad(out, table, cot)
    ref = torch.zeros((B, Nv, C), dtype=torch.float64, device=DEV)
    ref.scatter_add_(1, sel[:, :, None].expand(-1, -1, C), cot.double())
    assert float((grad.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    sets2 = cn.IndexSets(idx.clone(), Nv)
    (grad2,) = torch.autograd.grad(cn.gather_rows(table, sets2, set_of_b), table, cot)
    assert torch.equal(grad, grad2)


def test_fan_out_rows_for_several_readers():
    """control_nodes.fan_out: rows of stacked tensors handed to several readers each; the backward pass adds the readers' gradients of every row
    in one launch (gsr_multi_add) -- equal to autograd's own accumulation through plain indexing, zeros for rows nobody reads, five readers of
    one row (more than a launch item carries), a reader whose gradient is None."""
    g = torch.Generator(device="cpu").manual_seed(2)
    stacked = [torch.randn(shape, generator=g).to(DEV).requires_grad_(True) for shape in ((6, 300, 3), (6, 300, 4), (6, 300, 3))]
    plan = [(0, 0), (2, 0), (1, 0), (0, 0), (0, 1), (0, 0), (1, 3), (2, 5), (0, 0), (0, 0), (0, 0), (1, 3)]
    coef = [torch.randn((), generator=g).item() for _ in plan]
    unused = 4                                                    # (this reader contributes nothing to the loss)

    def loss_of(rows):
        return sum(c * (r * r).sum() for k, (c, r) in enumerate(zip(coef, rows)) if k != unused)

    rows = cn.fan_out(stacked, plan)
    assert all(torch.equal(r, stacked[a][i]) for r, (a, i) in zip(rows, plan))
    got = torch.autograd.grad(loss_of(rows), stacked)
    want = torch.autograd.grad(loss_of([stacked[a][i] for a, i in plan]), stacked)
    for a, (x, y) in enumerate(zip(got, want)):
        assert x.shape == y.shape and torch.allclose(x, y, rtol=1e-6, atol=1e-6), a
    assert float(got[0][2].abs().max()) == 0 and float(got[2][1].abs().max()) == 0        # rows without readers
    again = torch.autograd.grad(loss_of(cn.fan_out(stacked, plan)), stacked)
    assert all(torch.equal(x, y) for x, y in zip(got, again))


def test_node_blend_batch_packed_equals_the_per_attribute_call():
    """control_nodes.node_blend_batch_packed: the four node attributes as column ranges of one [B, M, 14] matrix (gsr_node_blend.attr_stride /
    grad_stride) -- bit-identical values and gradients to node_blend_batch on the four copies."""
    g = torch.Generator(device="cpu").manual_seed(8)
    n, m, B = 3000, 300, 5
    x = (torch.rand((n, 3), generator=g) - 0.5).to(DEV)
    mask = (torch.rand((n,), generator=g) > 0.2).float().to(DEV)
    nodes = (torch.rand((m, 3), generator=g) - 0.5).to(DEV)
    radius = (torch.randn((m,), generator=g) * 0.1 - 2.0).to(DEV).requires_grad_(True)
    weight = torch.randn((m, 1), generator=g).to(DEV).requires_grad_(True)
    attrs = (torch.randn((B, m, 14), generator=g) * 0.05).to(DEV).requires_grad_(True)
    cots = [torch.randn((B, n, c), generator=g).to(DEV) for c in (3, 4, 3)]

    def run(packed):
        if packed:
            out = cn.node_blend_batch_packed(x, mask, nodes, radius, weight, attrs, K=3)
        else:
            t, r, s, l = attrs.split([3, 4, 3, 4], -1)
            out = cn.node_blend_batch(x, mask, nodes, radius, weight, t, r, s, l, K=3)
        grads = torch.autograd.grad(out, (attrs, radius, weight), cots)
        return [o.detach() for o in out], grads

    (o1, g1), (o2, g2) = run(True), run(False)
    assert all(torch.equal(a, b) for a, b in zip(o1, o2))
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert float(g1[0].abs().sum()) > 0
