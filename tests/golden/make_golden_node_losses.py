#!/usr/bin/env python
"""Generates tests/golden/golden_node_losses.npz by running the reference's OWN node regularisers on seeded CPU tensors (imported from
/root/reference in the authoring container; fixtures are data only):
  * utils/deform_utils.py: cal_connectivity_from_points (:58-110, mode 'nn'), estimate_rotation (:130-166), cal_arap_error (:177-205);
  * utils/time_utils.py: ControlNodeWarp.arap_loss (:1128-1141) and elastic_loss (:1143-1165), with node_deform replaced by a recorded
    analytic motion so that the time samples the reference draws (torch.rand) and the node positions they produce are captured.
pytorch3d is not installed: knn_points is restated by brute force exactly as in make_golden_control_nodes.py."""
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
    return types.SimpleNamespace(dists=dist, idx=idx, knn=nn) if False else _KNN(dist, idx, nn)


class _KNN(tuple):
    """pytorch3d returns a namedtuple (dists, idx, knn): both attribute and positional access are used by the reference."""
    def __new__(cls, dists, idx, knn):
        return super().__new__(cls, (dists, idx, knn))
    dists = property(lambda s: s[0])
    idx = property(lambda s: s[1])
    knn = property(lambda s: s[2])


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


tk = _module("tkinter", W="w")
p3 = _module("pytorch3d")
p3.__path__ = []
p3.ops = _module("pytorch3d.ops", knn_points=knn_points, ball_query=None)
p3.io = _module("pytorch3d.io", load_ply=None)
p3.loss = _module("pytorch3d.loss")
p3.loss.__path__ = []
_module("pytorch3d.loss.mesh_laplacian_smoothing", cot_laplacian=None)
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.Tensor.cuda = lambda self, *a, **k: self
_to = torch.Tensor.to


def _to_cpu(self, *a, **k):          # the reference hard-codes device="cuda" (deform_utils.py:35-42)
    a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
    if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
        k["device"] = "cpu"
    return _to(self, *a, **k)


torch.Tensor.to = _to_cpu
import utils.deform_utils as D                                  # noqa: E402
import utils.time_utils as T                                    # noqa: E402

rng = np.random.default_rng(71)
out = {}

# ---- connectivity + rotations + ARAP error on a moving point set --------------------------------------------------------------------
M, Tn = 96, 4
base = torch.tensor(rng.uniform(-0.25, 0.25, size=(M, 3)).astype(np.float32))
ii, jj, nn, weight = D.cal_connectivity_from_points(base, K=10)
out["conn_points"], out["conn_ii"], out["conn_jj"], out["conn_nn"] = base.numpy(), ii.numpy(), jj.numpy(), nn.numpy()
seq = []
for k in range(Tn):      # a rotation about z growing with k + a non-rigid wobble
    a = 0.15 * k
    Rz = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
    seq.append(base @ Rz.T + 0.02 * k * torch.sin(7.0 * base[:, [1, 2, 0]]))
nodes_seq = torch.stack(seq).requires_grad_(True)             # [T, M, 3]
err = D.cal_arap_error(nodes_seq, ii, jj, nn)
err.backward()
out["arap_nodes_seq"], out["arap_error"], out["arap_grad"] = nodes_seq.detach().numpy(), err.detach().numpy(), nodes_seq.grad.numpy()
w1 = torch.zeros(M, 10)
w1[ii, nn] = 1
with torch.no_grad():
    R = D.estimate_rotation(nodes_seq[0].detach(), nodes_seq[2].detach(), ii, jj, nn, K=10, weight=w1, sample_idx=torch.arange(M))
    src = D.produce_edge_matrix_nfmt(nodes_seq[0].detach(), (M, 10, 3), ii, jj, nn, device="cpu")
    tgt = D.produce_edge_matrix_nfmt(nodes_seq[2].detach(), (M, 10, 3), ii, jj, nn, device="cpu")
    S = torch.bmm(src.permute(0, 2, 1), torch.bmm(torch.diag_embed(w1), tgt))
out["rot_S"], out["rot_R"] = S.numpy(), R.numpy()
# reflections: S whose nearest orthogonal matrix has det < 0 (the reference flips the column of the smallest singular value)
Sr = torch.tensor(rng.normal(size=(40, 3, 3)).astype(np.float32))
U, sig, V = torch.svd(Sr)
Rr = torch.bmm(V, U.permute(0, 2, 1))
flip = torch.nonzero(torch.det(Rr) <= 0).flatten()
Um = U.clone()
Um[flip, :, torch.argmin(sig[flip], dim=1)] *= -1
Rr[flip] = torch.bmm(V[flip], Um[flip].permute(0, 2, 1))
out["rot_S_random"], out["rot_R_random"], out["rot_n_reflections"] = Sr.numpy(), Rr.numpy(), np.int64(len(flip))

# ---- ControlNodeWarp.arap_loss / elastic_loss with a recorded motion --------------------------------------------------------------------
Mw = 64
w = T.ControlNodeWarp(is_blender=False, node_num=Mw, K=3, local_frame=True, d_rot_as_res=True, hyper_dim=0, with_arap_loss=False)
with torch.no_grad():
    w.nodes.copy_(torch.tensor(rng.uniform(-0.3, 0.3, size=(Mw, 3)).astype(np.float32)))
    w._node_radius.copy_(torch.tensor(np.log(rng.uniform(0.08, 0.3, size=Mw)).astype(np.float32)))
    w._node_weight.copy_(torch.tensor(rng.normal(size=(Mw, 1)).astype(np.float32)))
amp = torch.tensor(rng.normal(scale=0.05, size=(Mw, 3)).astype(np.float32))
phase = torch.tensor(rng.uniform(0, 6.28, size=(Mw, 3)).astype(np.float32))
out["warp_nodes"], out["warp_radius_raw"], out["warp_weight_raw"] = w.nodes.detach().numpy()[:, :3], w._node_radius.detach().numpy(), w._node_weight.detach().numpy()
out["motion_amp"], out["motion_phase"] = amp.numpy(), phase.numpy()
captured = {}


def node_deform(t, **kw):            # t: [M, T, 1]; d_xyz = amp * sin(9 t + phase): differentiable in amp (stand-in for the MLP's weights)
    captured["t"] = t.detach().clone()
    captured["d_xyz"] = amp_leaf[:, None, :] * torch.sin(9.0 * t + phase[:, None, :])
    captured["d_xyz"].retain_grad()
    return {"d_xyz": captured["d_xyz"]}


w.node_deform = node_deform
for name, call in (("arap4", lambda: w.arap_loss(t=torch.tensor([0.4]), delta_t=0.25, t_samp_num=4)),
                   ("arap2", lambda: w.arap_loss(t=torch.tensor([0.7]), delta_t=0.1)),
                   ("elastic", lambda: w.elastic_loss(t=torch.tensor([0.55]), delta_t=0.2))):
    torch.manual_seed(5)
    amp_leaf = amp.clone().requires_grad_(True)
    val = call()
    val.backward()
    out[name + "_t"], out[name + "_d_xyz"] = captured["t"][0, :, 0].numpy(), captured["d_xyz"].detach().numpy()
    out[name + "_value"], out[name + "_grad_d_xyz"], out[name + "_grad_amp"] = val.detach().numpy(), captured["d_xyz"].grad.numpy(), amp_leaf.grad.numpy()
    if name == "elastic":
        g_r = w._node_radius.grad
        out["elastic_grad_radius_raw"] = g_r.numpy().copy() if g_r is not None else np.zeros(Mw, np.float32)
# ---- DeformNetwork (time_utils.py:327-470) with the shipped flags, at width 32 to keep the fixture small -----------------------------------
torch.manual_seed(11)
net = T.DeformNetwork(is_blender=False, local_frame=True, W=32)
with torch.no_grad():                        # heads away from their near-zero initialisation, so that every output is exercised
    for head in (net.gaussian_warp, net.gaussian_scaling, net.gaussian_rotation, net.local_rotation):
        head.weight.normal_(std=0.2)
        head.bias.normal_(std=0.1)
for k, v in net.state_dict().items():
    out["net_" + k] = v.numpy()
xn = torch.tensor(rng.uniform(-0.5, 0.5, size=(37, 3)).astype(np.float32), requires_grad=True)
tn = torch.tensor(rng.uniform(0, 1, size=(37, 1)).astype(np.float32))
res = net(xn, tn)
(res["d_xyz"].sum() + 2 * res["d_rotation"].sum() + 3 * res["d_scaling"].sum() + 4 * res["local_rotation"].sum()).backward()
out["net_x"], out["net_t"] = xn.detach().numpy(), tn.numpy()
for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation"):
    out["net_out_" + k] = res[k].detach().numpy()
out["net_grad_first_layer"] = net.linear[0].weight.grad.numpy()
out["net_input_ch"], out["net_skips"] = np.int64(net.input_ch), np.asarray(net.skips)
# the default construction: initialisation statistics and sizes
torch.manual_seed(12)
full = T.DeformNetwork(is_blender=False, local_frame=True)
out["net_full_shapes"] = np.asarray([list(p.shape) + [0] * (2 - p.dim()) for p in full.state_dict().values()])
out["net_full_names"] = np.asarray(list(full.state_dict().keys()))
out["net_full_head_std"] = np.asarray([float(full.gaussian_warp.weight.std()), float(full.gaussian_scaling.weight.std()),
                                       float(full.gaussian_rotation.weight.std()), float(full.local_rotation.weight.std())])
np.savez_compressed(os.path.join(HERE, "golden_node_losses.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
