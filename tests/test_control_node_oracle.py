"""Pins oracle/control_node_oracle.py against golden vectors made by the reference's own ControlNodeWarp
(tests/golden/make_golden_control_nodes.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import control_node_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_control_nodes.npz"))


def rel(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-30))


def load(name, dtype=torch.float32, device="cpu"):
    M, N, K, local_frame, rot_res = [int(v) for v in G[f"{name}/flags"]]
    t = lambda k, rg=False: torch.tensor(G[f"{name}/{k}"], dtype=dtype, device=device).requires_grad_(rg)
    d = dict(x=t("x"), motion_mask=t("motion_mask"), nodes=t("nodes"), node_radius_raw=t("node_radius_raw", True),
             node_weight_raw=t("node_weight_raw", True), node_trans=t("node_d_xyz", True), node_rot=t("node_d_rotation", True),
             node_scale=t("node_d_scaling", True), local_rotation=t("node_local_rotation", True), K=K, local_frame=bool(local_frame),
             d_rot_as_res=bool(rot_res))
    return d


def check_against_golden(name, outs, d, nn=None, tol=2e-5):
    loss = 0
    for k, o in zip(("d_xyz", "d_rotation", "d_scaling"), outs):
        assert rel(o.detach().cpu(), G[f"{name}/out_{k}"]) < tol, (name, k)
        loss = loss + (o * torch.tensor(G[f"{name}/cot_{k}"], dtype=o.dtype, device=o.device)).sum()
    loss.backward()
    pairs = [("node_trans", "g_node_d_xyz"), ("node_rot", "g_node_d_rotation"), ("node_scale", "g_node_d_scaling"),
             ("node_radius_raw", "g_node_radius_raw"), ("node_weight_raw", "g_node_weight_raw")]
    if d["local_frame"]:
        pairs.append(("local_rotation", "g_node_local_rotation"))
    for key, gk in pairs:
        if float(np.abs(G[f"{name}/{gk}"]).max()) < 1e-5:          # analytically zero (K = 1: the single weight is u / u): rounding noise
            assert float(d[key].grad.abs().max()) < 1e-5, (name, key)
            continue
        assert rel(d[key].grad.cpu(), G[f"{name}/{gk}"]) < 10 * tol, (name, key, rel(d[key].grad.cpu(), G[f"{name}/{gk}"]))
    if nn is not None:
        w, dist, idx = nn
        assert np.array_equal(idx.cpu().numpy(), G[f"{name}/nn_idx"])
        assert rel(dist.cpu(), G[f"{name}/nn_dist"]) < 1e-6 and rel(w.detach().cpu(), G[f"{name}/nn_weight"]) < tol


@pytest.mark.parametrize("name", list(G["cases"]))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_matches_reference_control_node_warp(name, dtype):
    d = load(name, dtype)
    outs = O.node_blend(**d)
    nn = O.cal_nn_weight(d["x"], d["nodes"], d["node_radius_raw"], d["node_weight_raw"], d["K"])
    check_against_golden(name, outs, d, nn)


def test_knn_points_statement_on_trajectories():
    p = torch.tensor(G["knn_traj/p"][0])
    dist, idx = O.knn_points(p, p, 9)
    assert np.array_equal(idx.numpy(), G["knn_traj/idx"][0]) and np.allclose(dist.numpy(), G["knn_traj/dist"][0], rtol=1e-6, atol=1e-6)
    assert np.array_equal(idx[:, 0].numpy(), np.arange(200))      # the self-match comes first
