"""The SC-GS node regularisers of the dynamic mapping loop (slam/deform_model.py) against fixtures recorded from the reference's own
utils/deform_utils.py (cal_connectivity_from_points, estimate_rotation, cal_arap_error) and ControlNodeWarp.arap_loss / elastic_loss
(tests/golden/make_golden_node_losses.py). CPU: the tensor programs with the rotation solver replaced by torch.svd (the product path has
no CPU rotation solver); GPU: the product path (HIP k-NN, gsr_kabsch_rotations, HIP RBF weights)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "4dgs-slam_amd"))
G = np.load(os.path.join(HERE, "golden", "golden_node_losses.npz"))


def _svd_rotations(S):
    """Test stand-in for gsr_kabsch_rotations on CPU: the same definition through torch.svd."""
    shape, S = S.shape, S.reshape(-1, 3, 3)
    U, sig, V = torch.svd(S)
    R = V @ U.transpose(-1, -2)
    flip = torch.det(R) <= 0
    if flip.any():
        Um = U.clone()
        cols = torch.argmin(sig[flip], dim=-1)
        idx = torch.nonzero(flip, as_tuple=False).flatten()
        Um[idx, :, cols] *= -1
        R[idx] = V[idx] @ Um[idx].transpose(-1, -2)
    return R.reshape(shape)


def _dense(ii, jj, nn, Nv, K=10):
    nn_idx = torch.zeros((Nv, K), dtype=torch.int64)
    keep = torch.zeros((Nv, K), dtype=torch.bool)
    nn_idx[ii, nn] = torch.as_tensor(jj)
    keep[ii, nn] = True
    return nn_idx, keep


def test_arap_error_and_its_gradient_match_the_reference():
    from slam.deform_model import arap_error
    seq = torch.tensor(G["arap_nodes_seq"], requires_grad=True)
    nn_idx, keep = _dense(G["conn_ii"], G["conn_jj"], G["conn_nn"], seq.shape[1])
    err = arap_error(seq, nn_idx, keep, rotations=_svd_rotations)
    err.backward()
    assert abs(float(err.detach()) - float(G["arap_error"])) <= 2e-5 * abs(float(G["arap_error"]))
    np.testing.assert_allclose(seq.grad.numpy(), G["arap_grad"], rtol=2e-4, atol=2e-6)
    # batched form: two copies give the same numbers per leading index
    both = arap_error(torch.stack([seq.detach(), seq.detach()]), torch.stack([nn_idx, nn_idx]), torch.stack([keep, keep]), rotations=_svd_rotations)
    assert both.shape == (2,) and torch.allclose(both, err.detach().expand(2), rtol=1e-6)


def test_estimate_rotation_matches_the_reference():
    from slam.deform_model import edge_matrix, estimate_rotation
    seq = torch.tensor(G["arap_nodes_seq"])
    nn_idx, keep = _dense(G["conn_ii"], G["conn_jj"], G["conn_nn"], seq.shape[1])
    E0, E2 = edge_matrix(seq[0], nn_idx, keep), edge_matrix(seq[2], nn_idx, keep)
    S = torch.einsum("nka,nk,nkb->nab", E0, keep.float(), E2)
    np.testing.assert_allclose(S.numpy(), G["rot_S"], rtol=1e-5, atol=1e-7)
    R = estimate_rotation(E0, E2, keep.float(), rotations=_svd_rotations)
    np.testing.assert_allclose(R.numpy(), G["rot_R"], atol=2e-5)
    assert int(G["rot_n_reflections"]) > 0
    np.testing.assert_allclose(_svd_rotations(torch.tensor(G["rot_S_random"])).numpy(), G["rot_R_random"], atol=2e-5)


def _motion(times, amp):
    t = torch.tensor(np.asarray(times, np.float32))[None, :, None]
    return amp[:, None, :] * torch.sin(9.0 * t + torch.tensor(G["motion_phase"])[:, None, :])


def test_elastic_error_matches_the_reference_given_its_neighbours():
    """elastic_loss' own expression on the recorded motion; the RBF neighbour weights are the GPU test's business (they come from the HIP
    cal_nn_weight, golden-tested on its own): here they are rebuilt with the reference's formula (time_utils.py:1000-1011)."""
    from slam.deform_model import elastic_error
    nodes = torch.tensor(G["warp_nodes"])
    d = ((nodes[:, None] - nodes[None]) ** 2).sum(-1)
    nn_dist, nn_idx = torch.topk(d, 3, dim=-1, largest=False, sorted=True)
    radius, nw = torch.exp(torch.tensor(G["warp_radius_raw"])), torch.sigmoid(torch.tensor(G["warp_weight_raw"]))
    w = torch.exp(-nn_dist / (2 * radius[nn_idx] ** 2)) * nw[nn_idx][..., 0] + 1e-7
    w = w / w.sum(-1, keepdim=True)
    amp = torch.tensor(G["motion_amp"], requires_grad=True)
    nodes_t = nodes[:, None, :] + _motion(G["elastic_t"], amp)
    np.testing.assert_allclose(_motion(G["elastic_t"], amp).detach().numpy(), G["elastic_d_xyz"], atol=1e-6)
    val = elastic_error(nodes_t, w[:, 1:], nn_idx[:, 1:])
    val.backward()
    assert abs(float(val) - float(G["elastic_value"])) <= 1e-4 * abs(float(G["elastic_value"]))
    np.testing.assert_allclose(amp.grad.numpy(), G["elastic_grad_amp"], rtol=2e-3, atol=1e-5)


def test_loss_time_samples_follow_the_reference_draws():
    from slam.deform_model import draw_loss_times
    torch.manual_seed(5)
    plan = draw_loss_times(0.4, 0.25, 4, 0.2)
    np.testing.assert_allclose(plan["arap"], G["arap4_t"], atol=1e-6)          # the reference's arap_loss(t=0.4, delta_t=0.25, t_samp_num=4) after manual_seed(5)
    assert len(plan["elastic"]) == 8 and all(0.4 - 0.2 <= x <= 0.4 + 0.2 for x in plan["elastic"])
    torch.manual_seed(5)
    te = 0.55 + 0.2 * (float(torch.rand(())) - 0.5)
    np.testing.assert_allclose((torch.rand(8) * 0.2 + te - 0.1).numpy(), G["elastic_t"], atol=1e-6)


@pytest.mark.gpu
def test_kabsch_kernel_matches_the_reference_rotations():
    from slam.deform_model import kabsch_rotations
    for a, b in (("rot_S", "rot_R"), ("rot_S_random", "rot_R_random")):
        R = kabsch_rotations(torch.tensor(G[a], device="cuda"))
        np.testing.assert_allclose(R.cpu().numpy(), G[b], atol=2e-5)
    Z = kabsch_rotations(torch.zeros((3, 3, 3), device="cuda"))
    assert torch.equal(Z, torch.eye(3, device="cuda").expand(3, 3, 3))
    R = kabsch_rotations(torch.randn(4, 5, 3, 3, device="cuda"))                      # leading dimensions; proper rotations
    assert R.shape == (4, 5, 3, 3) and torch.allclose(torch.det(R), torch.ones(4, 5, device="cuda"), atol=1e-5)
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3, device="cuda").expand(4, 5, 3, 3), atol=1e-5)


@pytest.mark.gpu
def test_node_regularisers_product_path_matches_the_reference():
    """ControlNodes.arap_loss / elastic_loss through the HIP k-NN, the HIP rotations and the HIP RBF weights, on the recorded motion and
    the recorded time samples: values and gradients of the reference's ControlNodeWarp.arap_loss / elastic_loss."""
    from slam.deform_model import ControlNodes, connectivity_from_points
    pts = torch.tensor(G["conn_points"], device="cuda")
    nn_idx, keep = connectivity_from_points(pts, K=10)
    want_idx, want_keep = _dense(G["conn_ii"], G["conn_jj"], G["conn_nn"], pts.shape[0])
    assert torch.equal(keep.cpu(), want_keep) and torch.equal(nn_idx.cpu()[want_keep], want_idx[want_keep])
    cn = ControlNodes(node_num=64, device="cuda")
    cn.nodes = torch.nn.Parameter(torch.tensor(G["warp_nodes"], device="cuda"))
    cn._node_radius = torch.nn.Parameter(torch.tensor(G["warp_radius_raw"], device="cuda"))
    cn._node_weight = torch.nn.Parameter(torch.tensor(G["warp_weight_raw"], device="cuda"))
    phase = torch.tensor(G["motion_phase"], device="cuda")
    for name, fn in (("arap4", cn.arap_loss), ("arap2", cn.arap_loss), ("elastic", cn.elastic_loss)):
        amp = torch.tensor(G["motion_amp"], device="cuda", requires_grad=True)
        cn.node_deform = lambda t, key=None, amp=amp: {"d_xyz": amp * torch.sin(9.0 * t + phase)}      # the recorded stand-in for the node MLP
        val = fn([float(x) for x in G[name + "_t"]])
        val.backward()
        assert abs(float(val.detach()) - float(G[name + "_value"])) <= 3e-4 * abs(float(G[name + "_value"])), (name, float(val.detach()), float(G[name + "_value"]))
        np.testing.assert_allclose(amp.grad.cpu().numpy(), G[name + "_grad_amp"], rtol=5e-3, atol=2e-5 * np.abs(G[name + "_grad_amp"]).max())
        if name == "elastic":
            np.testing.assert_allclose(cn._node_radius.grad.cpu().numpy(), G["elastic_grad_radius_raw"], rtol=5e-3, atol=1e-6)


def test_node_network_is_the_reference_deform_network():
    """slam.deform_model.NodeNetwork against the reference's DeformNetwork(is_blender=False, local_frame=True): the reference's state_dict
    loads by name, outputs and a weight gradient match; the default construction has the reference's layer shapes and head
    initialisation scales."""
    from slam.deform_model import NodeNetwork
    net = NodeNetwork(W=32)
    sd = {k[4:]: torch.tensor(G[k]) for k in G.files if k.startswith("net_") and "." in k}
    net.load_state_dict(sd, strict=True)
    assert net.input_ch == int(G["net_input_ch"]) and list(net.skips) == list(G["net_skips"])
    out = net(torch.tensor(G["net_x"]), torch.tensor(G["net_t"]))
    for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation"):
        np.testing.assert_allclose(out[k].detach().numpy(), G["net_out_" + k], rtol=2e-5, atol=2e-6)
    (out["d_xyz"].sum() + 2 * out["d_rotation"].sum() + 3 * out["d_scaling"].sum() + 4 * out["local_rotation"].sum()).backward()
    np.testing.assert_allclose(net.linear[0].weight.grad.numpy(), G["net_grad_first_layer"], rtol=2e-4, atol=2e-6)
    torch.manual_seed(12)
    full = NodeNetwork()
    names = list(full.state_dict().keys())
    assert names == [str(n) for n in G["net_full_names"]]
    shapes = [list(p.shape) + [0] * (2 - p.dim()) for p in full.state_dict().values()]
    assert shapes == G["net_full_shapes"].tolist()
    std = [float(full.gaussian_warp.weight.std()), float(full.gaussian_scaling.weight.std()), float(full.gaussian_rotation.weight.std()),
           float(full.local_rotation.weight.std())]
    np.testing.assert_allclose(std, G["net_full_head_std"], rtol=0.2)


def test_node_initialisation_rules():
    """ControlNodeWarp.init (:904-951) / extend_node (:953-981): all points while fewer than the budget, radius log(0.1 * range + 1e-7)."""
    from slam.deform_model import ControlNodes, DeformModel
    pts = torch.tensor(np.random.default_rng(3).uniform(-0.4, 0.7, size=(40, 3)).astype(np.float32))
    cn = ControlNodes(node_num=64, device="cpu", W=16)
    cn.init(pts)
    assert cn.node_num == 40 and torch.equal(cn.nodes.detach(), pts)
    want = float(torch.log(0.1 * (pts.max() - pts.min()) + 1e-7))
    assert torch.allclose(cn._node_radius.detach(), torch.full((40,), want)) and float(cn._node_weight.abs().sum()) == 0.0
    dm = DeformModel(node_num=64, device="cpu")
    dm.deform = ControlNodes(node_num=64, device="cpu", W=16)
    dm.extend_node_from_point(pts)
    assert abs(dm.lr - 0.00016 * 5) < 1e-12 and [g["name"] for g in dm.optimizer.param_groups] == ["deform", "nodes"]
    more = pts[:7] + 0.05
    dm.extend_node_from_point(more)                                     # fewer new points than nodes: all of them are appended
    assert dm.deform.node_num == 47 and dm.deform._node_radius.shape == (47,) and dm.deform._node_weight.shape == (47, 1)
    assert dm.optimizer.param_groups[1]["params"][0] is dm.deform.nodes


def test_batched_node_network_evaluation_equals_the_direct_one():
    """ControlNodes.begin_iteration: all time samples of an iteration through the network as ONE batch; every later node_deform(t, key)
    lookup must return what the direct evaluation at that time returns, and the lookups must vanish at end_iteration."""
    from slam.deform_model import ControlNodes
    torch.manual_seed(0)
    cn = ControlNodes(node_num=32, device="cpu", W=16)
    cn.init(torch.randn(20, 3) * 0.3)
    with torch.no_grad():
        for head in (cn.network.gaussian_warp, cn.network.gaussian_rotation, cn.network.gaussian_scaling, cn.network.local_rotation):
            head.weight.normal_(std=0.3)
    times = [0.1, 0.25, 0.1 + 0.05, 0.9]
    cn.begin_iteration(times + [0.25])                       # duplicates collapse
    assert len(cn._batch) == 4
    for tv in times:
        a = cn.node_deform(torch.full((cn.node_num, 1), tv), tv)
        b = cn.network(cn.nodes.detach(), torch.full((cn.node_num, 1), tv))
        for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation"):
            assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6), k
    pos = cn.node_positions(times)
    assert pos.shape == (cn.node_num, 4, 3)
    cn.end_iteration()
    assert cn._batch is None
    # samples of which only the positions are read (the regularisers'): the trunk and the translation head see them, the other heads
    # do not; asking for every head of such a sample falls back to the direct evaluation; values and gradients are those of separate calls
    extra = [0.4, 0.55, 0.1]                                  # 0.1 is also a full sample
    cn.begin_iteration(times[:2], positions_only=extra)
    assert set(cn._batch) == {0.1, 0.25, 0.4, 0.55} and set(cn._batch[0.4]) == {"d_xyz"} and "d_rotation" in cn._batch[0.1]
    pos = cn.node_positions([0.4, 0.55, 0.1])
    full = cn.node_deform(torch.full((cn.node_num, 1), 0.4), 0.4)
    loss = pos.square().sum() + cn.node_deform(None, 0.25)["d_rotation"].sum() + full["d_scaling"].sum()
    loss.backward()
    got = [None if p.grad is None else p.grad.clone() for p in cn.network.parameters()]
    cn.end_iteration()
    cn.network.zero_grad()
    direct = lambda tv: cn.network(cn.nodes.detach(), torch.full((cn.node_num, 1), tv))
    want_pos = cn.nodes.detach()[:, None, :] + torch.stack([direct(tv)["d_xyz"] for tv in (0.4, 0.55, 0.1)], 1)
    assert torch.allclose(pos, want_pos, rtol=1e-5, atol=1e-6)
    (want_pos.square().sum() + direct(0.25)["d_rotation"].sum() + direct(0.4)["d_scaling"].sum()).backward()
    for a, b in zip(got, cn.network.parameters()):
        # (sums over ~2 000 rows in a different order: the absolute slack scales with the gradient's magnitude -- a host with another BLAS
        # threading, e.g. the GPU box's, needs it)
        assert (a is None and b.grad is None) or torch.allclose(a, b.grad, rtol=1e-4, atol=1e-6 + 2e-5 * float(b.grad.abs().max()))


@pytest.mark.gpu
def test_batched_node_network_on_the_device_goes_through_the_fused_trunk_and_equals_the_layerwise_route(monkeypatch):
    """ControlNodes.begin_iteration on the device = begin_iteration_indexed (the whole network as ONE autograd node on the dense kernels, round 6)
    with the bookkeeping by host time on top; GSR_BATCH_TRUNK=0 keeps the layer-by-layer library route. Same lookups (full samples, position-only
    samples, the blended rows of the Gaussians), same values and gradients to fp32-GEMM accuracy."""
    from slam.deform_model import ControlNodes
    results = {}
    for route in ("1", "0"):
        monkeypatch.setenv("GSR_BATCH_TRUNK", route)
        torch.manual_seed(0)
        cn = ControlNodes(node_num=256, device="cuda")
        cn.init((torch.randn(200, 3) * 0.3).cuda())
        with torch.no_grad():
            for head in (cn.network.gaussian_warp, cn.network.gaussian_rotation, cn.network.gaussian_scaling, cn.network.local_rotation):
                head.weight.normal_(std=0.3)
        x = (torch.randn(3000, 3, generator=torch.Generator().manual_seed(1)) * 0.3).cuda()
        times, extra = [0.1, 0.25, 0.9], [0.4, 0.55, 0.1]
        cn.begin_iteration(times, positions_only=extra, blend=(x, None))
        assert set(cn._batch) == {0.1, 0.25, 0.9, 0.4, 0.55} and set(cn._batch[0.4]) == {"d_xyz"} and "d_rotation" in cn._batch[0.1]
        pos = cn.node_positions(extra)
        warped = [cn.forward(x, torch.full((x.shape[0], 1), tv, device="cuda"), t_key=tv) for tv in times]
        loss = pos.square().sum() + sum((w["d_xyz"].square().sum() + w["d_rotation"].sum() + w["d_scaling"].square().sum()) for w in warped)
        loss.backward()
        results[route] = (pos.detach().clone(), [w["d_xyz"].detach().clone() for w in warped],
                          [None if p.grad is None else p.grad.clone() for p in cn.network.parameters()])
        cn.end_iteration()
        assert cn._batch is None
    (pa, wa, ga), (pb, wb, gb) = results["1"], results["0"]
    # (eight 256-wide layers in fp32 on two GEMM implementations: differences of a few 1e-6 of the largest entry)
    assert torch.allclose(pa, pb, rtol=1e-5, atol=4e-6 * float(pb.abs().max()))
    for a, b in zip(wa, wb):
        assert torch.allclose(a, b, rtol=1e-5, atol=4e-6 * float(b.abs().max()))
    assert sum(g is not None for g in ga) >= 18
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 + 2e-5 * float(b.abs().max()))


@pytest.mark.gpu
def test_fused_regularisers_equal_the_op_by_op_ones():
    """arap_error / elastic_error on the device through the one-launch kernels (gsr_arap_forward / _backward, gsr_elastic_forward / _backward)
    against the tensor programs they replace (the ones the CPU tests compare with the reference): several views, values and gradients."""
    from slam import deform_model as dm
    g = torch.Generator().manual_seed(5)
    V, T, M = 3, 4, 200
    base = torch.randn(M, 3, generator=g) * 0.3
    motion = (torch.randn(V, T, M, 3, generator=g) * 0.02).cuda()
    weights = torch.rand(M, 2, generator=g).cuda().requires_grad_(True)
    results = {}
    for fused in (True, False):
        dm.FUSED_REGULARISERS = fused
        try:
            d = motion.clone().requires_grad_(True)
            seq = base.cuda() + d
            nn_idx, keep = dm.connectivity_from_points(seq[:, 0], K=10, radius=0.25)
            arap = dm.arap_error(seq, nn_idx, keep)
            knn = dm.control_nodes.knn_points(base.cuda()[None], base.cuda()[None], K=3).idx[0, :, 1:]
            elastic = dm.elastic_error(seq.permute(0, 2, 1, 3), weights, knn)
            (arap * torch.tensor([1.0, 0.5, 2.0], device="cuda")).sum().backward(retain_graph=True)
            ga = d.grad.clone(); d.grad = None
            (elastic * torch.tensor([1.0, 0.5, 2.0], device="cuda")).sum().backward()
            results[fused] = (arap.detach(), elastic.detach(), ga, d.grad.clone(), weights.grad.clone())
            weights.grad = None
        finally:
            dm.FUSED_REGULARISERS = True
    assert bool(keep.any()) and not bool(keep.all())
    for name, a, b in zip(("arap", "elastic", "d arap", "d elastic", "d weights"), results[True], results[False]):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-4, atol=2e-6 * float(b.abs().max())), (name, float((a - b).abs().max()), float(b.abs().max()))


def test_row_groups_and_the_one_layer_heads_on_cpu():
    """Host logic of the fused node network: the row-group choice of the batched weight-gradient GEMMs (a divisor of the row count with about
    2 000 rows per group, else one group) and, on CPU tensors, heads_from_embedding = the four head layers applied to the op-by-op trunk."""
    from slam.deform_model import NodeNetwork, _row_groups
    for rows, want in ((33280, 16), (69632, 34), (6000, 3), (6007, 1), (512, 1), (71680, 35)):
        g = _row_groups(rows)
        assert g == want and rows % g == 0 and (g == 1 or 1024 <= rows // g <= 4096), (rows, g)
    torch.manual_seed(1)
    net = NodeNetwork()
    for _, head in net.heads():
        torch.nn.init.normal_(head.weight, std=0.05)
    emb = torch.randn(50, net.input_ch)
    out = net.heads_from_embedding(emb)
    h = net.trunk(emb)
    want = torch.cat([m(h) for _, m in net.heads()], -1)
    assert out.shape == (50, 14) and torch.allclose(out, want, atol=1e-6)
    d = net.from_embedding(emb)
    assert torch.allclose(d["d_xyz"], want[:, :3], atol=1e-6) and torch.allclose(d["local_rotation"], want[:, 10:], atol=1e-6)
