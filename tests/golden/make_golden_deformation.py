#!/usr/bin/env python
"""Generates tests/golden/golden_deformation.npz by calling the reference's OWN HexPlane field and deformation network
(gaussian_splatting/utils/hexplane.py: HexPlaneField; utils/deformation.py: deform_network -- imported from /root/reference,
authoring container only) on seeded CPU tensors.  Fixtures are data only: parameters, inputs, outputs and autograd gradients.

Two groups of cases:
  field/*   HexPlaneField with deliberately unequal resolutions (x 6, y 5, z 4, t 3; two levels) so that a swapped axis, a
            swapped plane or a wrong level order cannot cancel.  Points inside, outside (clamped by normalize_aabb) and exactly
            on texel centres / the border; times inside and outside [-1, 1] (time is NOT clamped by normalize_aabb, only by the
            sampler's border padding).
  net1/*, net2/*   the full deform_network for defor_depth 1 (the shipped default, arguments/__init__.py:78) and 2, on a
            reduced plane resolution so that the state dict stays small."""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
tk = types.ModuleType("tkinter")
tk.W = "w"                                                  # utils/deformation.py:5 has a stray `from tkinter import W`
sys.modules["tkinter"] = tk
from arguments import ModelHiddenParams                      # noqa: E402
from gaussian_splatting.utils.hexplane import HexPlaneField  # noqa: E402
import utils.deformation as ref_deformation                 # noqa: E402

rng = np.random.default_rng(11)
out = {}

# ---------------------------------------------------------------- field ----------------------------------------------------
cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [6, 5, 4, 3]}
field = HexPlaneField(1.6, cfg, [1, 2])
with torch.no_grad():
    for level in field.grids:
        for plane in level:
            plane.copy_(torch.tensor(rng.uniform(0.1, 1.5, size=tuple(plane.shape)).astype(np.float32)))
n = 256
pts = rng.uniform(-2.0, 2.0, size=(n, 3)).astype(np.float32)
tim = rng.uniform(-1.3, 1.3, size=(n, 1)).astype(np.float32)
# exact texel centres of level 0 (normalised coordinate -1 + 2 i / (res - 1); normalize_aabb maps p -> -p / 1.6), the aabb
# faces and the time-plane rows
for i in range(6):
    pts[i] = [-1.6 * (-1 + 2 * i / 5), -1.6 * (-1 + 2 * (i % 5) / 4), -1.6 * (-1 + 2 * (i % 4) / 3)]
    tim[i] = -1 + 2 * (i % 3) / 2
pts[6], pts[7] = [1.6, -1.6, 1.6], [-1.6, 1.6, 0.0]
tim[6], tim[7] = 1.0, -1.0
pts_t = torch.tensor(pts, requires_grad=True)
feat = field(pts_t, torch.tensor(tim))
cot = torch.tensor(rng.normal(size=tuple(feat.shape)).astype(np.float32))
(feat * cot).sum().backward()
out["field/resolution"] = np.array(cfg["resolution"])
out["field/multires"] = np.array([1, 2])
out["field/aabb"] = field.aabb.detach().numpy()
out["field/pts"], out["field/time"] = pts, tim
out["field/features"] = feat.detach().numpy()
out["field/cotangent"] = cot.numpy()
out["field/g_pts"] = pts_t.grad.numpy()
for li, level in enumerate(field.grids):
    for pi, plane in enumerate(level):
        out[f"field/plane_{li}_{pi}"] = plane.detach().numpy()
        out[f"field/g_plane_{li}_{pi}"] = plane.grad.numpy()


# ---------------------------------------------------------------- full network ----------------------------------------------
def hidden_params(**over):
    parser = argparse.ArgumentParser()
    args = ModelHiddenParams(parser).extract(parser.parse_args([]))
    for k, v in over.items():
        setattr(args, k, v)
    return args


for tag, depth in (("net1", 1), ("net2", 2)):
    torch.manual_seed(5 + depth)
    args = hidden_params(defor_depth=depth, multires=[1, 2],
                         kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                         "resolution": [8, 7, 6, 5]})
    net = ref_deformation.deform_network(args, "cpu")
    with torch.no_grad():                                   # time planes start as ones (hexplane.py:70-71): make them informative
        for name, p in net.named_parameters():
            if "grid" in name:
                p.copy_(torch.tensor(rng.uniform(0.2, 1.2, size=tuple(p.shape)).astype(np.float32)))
    n = 96
    point = torch.tensor(rng.uniform(-1.8, 1.8, size=(n, 3)).astype(np.float32), requires_grad=True)
    scales = torch.tensor(rng.normal(-4.0, 0.5, size=(n, 3)).astype(np.float32), requires_grad=True)
    rots = torch.tensor(rng.normal(size=(n, 4)).astype(np.float32), requires_grad=True)
    opac = torch.tensor(rng.normal(size=(n, 1)).astype(np.float32))
    shs = torch.tensor(rng.normal(size=(n, 16, 3)).astype(np.float32))
    time = torch.tensor(np.full((n, 1), 0.37, np.float32))
    outs = net(point, scales, rots, opac, shs, time)
    names = ["means3D", "scales", "rotations", "dx", "ds", "dr"]
    loss = 0
    for name, o in zip(names, outs):
        c = torch.tensor(rng.normal(size=tuple(o.shape)).astype(np.float32))
        out[f"{tag}/out_{name}"], out[f"{tag}/cot_{name}"] = o.detach().numpy(), c.numpy()
        loss = loss + (o * c).sum()
    loss.backward()
    out[f"{tag}/resolution"], out[f"{tag}/multires"], out[f"{tag}/defor_depth"] = np.array([8, 7, 6, 5]), np.array([1, 2]), depth
    for k, v in dict(point=point, scales=scales, rotations=rots, opacity=opac, shs=shs, time=time).items():
        out[f"{tag}/in_{k}"] = v.detach().numpy()
        if v.grad is not None:
            out[f"{tag}/g_in_{k}"] = v.grad.numpy()
    keys = []
    for k, v in net.state_dict().items():
        out[f"{tag}/state/{k}"] = v.numpy()
        keys.append(k)
    out[f"{tag}/state_keys"] = np.array(keys)
    for k, p in net.named_parameters():
        out[f"{tag}/grad/{k}"] = (p.grad if p.grad is not None else torch.zeros(0)).numpy()
    out[f"{tag}/mlp_parameter_count"] = len(net.get_mlp_parameters())
    out[f"{tag}/grid_parameter_count"] = len(net.get_grid_parameters())

# ---------------------------------------------------------------- default geometry (shapes only) ----------------------------
net = ref_deformation.deform_network(hidden_params(), "cpu")
out["default/state_keys"] = np.array(list(net.state_dict().keys()))
out["default/state_shapes"] = np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])
out["default/parameter_count"] = sum(p.numel() for p in net.parameters())

np.savez_compressed(os.path.join(HERE, "golden_deformation.npz"), **out)
print("wrote", os.path.join(HERE, "golden_deformation.npz"), os.path.getsize(os.path.join(HERE, "golden_deformation.npz")), "bytes")
