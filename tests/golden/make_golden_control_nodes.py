#!/usr/bin/env python
"""Generates tests/golden/golden_control_nodes.npz by running the reference's OWN SC-GS control-node warp
(utils/time_utils.py: ControlNodeWarp.cal_nn_weight :981-1015 and ControlNodeWarp.forward :1192-1296, imported from
/root/reference in the authoring container) on seeded CPU tensors.  Fixtures are data only.

pytorch3d is a dependency of that file that is not vendored and not installed (requirements.txt:19, unpinned git install);
the only routine of it on this path is pytorch3d.ops.knn_points, whose published semantics -- for every point of p1 the K
nearest points of p2 by SQUARED Euclidean distance, sorted ascending, int64 indices -- are restated in the stub below
(brute force).  Everything else (the RBF weights, the local-frame blend, the residual rotation / scale, the motion mask, the
gradients) is the reference's code."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    dist, idx = torch.topk(d, K, dim=-1, largest=False, sorted=True)
    nn = torch.gather(p2[:, None].expand(-1, p1.shape[1], -1, -1), 2, idx[..., None].expand(-1, -1, -1, p2.shape[-1])) if return_nn else None
    return dist, idx, nn


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


p3 = _module("pytorch3d")
p3.__path__ = []
p3.ops = _module("pytorch3d.ops", knn_points=knn_points, ball_query=None)
p3.io = _module("pytorch3d.io", load_ply=None)
p3.loss = _module("pytorch3d.loss")
p3.loss.__path__ = []
_module("pytorch3d.loss.mesh_laplacian_smoothing", cot_laplacian=None)
torch.nn.Module.cuda = lambda self, *a, **k: self            # the reference hard-codes .cuda() (time_utils.py:822)
torch.Tensor.cuda = lambda self, *a, **k: self
import utils.time_utils as T                                   # noqa: E402

rng = np.random.default_rng(23)
out = {}
cases = {
    # the shipped flags (arguments/__init__.py: K=3, node_num=512, local_frame=True, d_rot_as_res=True, hyper_dim=0)
    "shipped": dict(M=512, N=1500, K=3, local_frame=True, d_rot_as_res=True),
    "global_frame": dict(M=64, N=300, K=3, local_frame=False, d_rot_as_res=True),
    "absolute_rotation": dict(M=64, N=300, K=4, local_frame=True, d_rot_as_res=False),
    "k1": dict(M=17, N=50, K=1, local_frame=False, d_rot_as_res=True),
}
out["cases"] = np.array(list(cases))
for name, c in cases.items():
    torch.manual_seed(3)
    M, N, K = c["M"], c["N"], c["K"]
    w = T.ControlNodeWarp(is_blender=False, node_num=M, K=K, local_frame=c["local_frame"], d_rot_as_res=c["d_rot_as_res"], hyper_dim=0,
                          with_arap_loss=False)
    with torch.no_grad():
        w.nodes.copy_(torch.tensor(rng.uniform(-1, 1, size=(M, 3)).astype(np.float32)))
        w._node_radius.copy_(torch.tensor(np.log(rng.uniform(0.15, 0.6, size=M)).astype(np.float32)))
        w._node_weight.copy_(torch.tensor(rng.normal(size=(M, 1)).astype(np.float32)))
    # per-node attributes as leaves (in the reference they come out of the node MLP, time_utils.py:1038-1050); replacing
    # node_deform keeps forward() itself untouched
    attrs = {"d_xyz": torch.tensor(rng.normal(scale=0.1, size=(M, 3)).astype(np.float32), requires_grad=True),
             "d_rotation": torch.tensor(rng.normal(scale=0.2, size=(M, 4)).astype(np.float32), requires_grad=True),
             "d_scaling": torch.tensor(rng.normal(scale=0.1, size=(M, 3)).astype(np.float32), requires_grad=True),
             "local_rotation": torch.tensor(rng.normal(scale=0.3, size=(M, 4)).astype(np.float32), requires_grad=True),
             "d_opacity": None, "d_color": None}
    w.node_deform = lambda t, **kw: attrs
    x = torch.tensor(rng.uniform(-1.2, 1.2, size=(N, 3)).astype(np.float32))
    mask = torch.tensor((rng.uniform(size=(N, 1)) < 0.8).astype(np.float32) * rng.uniform(0.5, 1.0, size=(N, 1)).astype(np.float32))
    w.eval()
    res = w(x, torch.tensor(0.3), None, mask)
    nn_weight, nn_dist, nn_idx = w.cal_nn_weight(x=x, feature=None)
    loss = 0
    for k in ("d_xyz", "d_rotation", "d_scaling"):
        cot = torch.tensor(rng.normal(size=tuple(res[k].shape)).astype(np.float32))
        out[f"{name}/out_{k}"], out[f"{name}/cot_{k}"] = res[k].detach().numpy(), cot.numpy()
        loss = loss + (res[k] * cot).sum()
    loss.backward()
    out[f"{name}/flags"] = np.array([M, N, K, int(c["local_frame"]), int(c["d_rot_as_res"])])
    out[f"{name}/x"], out[f"{name}/motion_mask"] = x.numpy(), mask.numpy()
    out[f"{name}/nodes"] = w.nodes.detach().numpy()
    out[f"{name}/node_radius_raw"], out[f"{name}/node_weight_raw"] = w._node_radius.detach().numpy(), w._node_weight.detach().numpy()
    out[f"{name}/nn_weight"], out[f"{name}/nn_dist"], out[f"{name}/nn_idx"] = nn_weight.detach().numpy(), nn_dist.numpy(), nn_idx.numpy()
    for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation"):
        out[f"{name}/node_{k}"] = attrs[k].detach().numpy()
        out[f"{name}/g_node_{k}"] = attrs[k].grad.numpy() if attrs[k].grad is not None else np.zeros(0, np.float32)
    out[f"{name}/g_node_radius_raw"] = w._node_radius.grad.numpy()
    out[f"{name}/g_node_weight_raw"] = w._node_weight.grad.numpy()
    assert w.nodes.grad is None                                # nodes are detached on this path (:993)

# knn_points call shapes the reference uses elsewhere (trajectories: D = 3 * t_samp, K + 1 with self-match first)
traj = torch.tensor(rng.normal(size=(1, 200, 24)).astype(np.float32))
d, i, _ = knn_points(traj, traj, K=9)
out["knn_traj/p"], out["knn_traj/dist"], out["knn_traj/idx"] = traj.numpy(), d.numpy(), i.numpy()

np.savez_compressed(os.path.join(HERE, "golden_control_nodes.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "golden_control_nodes.npz")), "bytes")
