#!/usr/bin/env python
"""Development: norm-relative gradient errors of the fused trunk (dense layers on / off) and of the op-by-op fp32 network against fp64 autograd."""
import os, sys, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam import deform_model as dm
DEV = "cuda"
torch.manual_seed(3)
net = dm.NodeNetwork().to(DEV)
for _, head in net.heads():
    torch.nn.init.normal_(head.weight, std=0.05)
names = [k for k, _ in net.named_parameters()]
def grads64(emb, cot):
    n64 = dm.NodeNetwork().to(DEV).double()
    n64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    h = n64.trunk(emb.double())
    out = torch.cat([m(h) for _, m in n64.heads()], -1)
    out.backward(cot.double())
    return out.detach(), {k: p.grad for k, p in n64.named_parameters()}
for rows in (6000, 33280, 6007):
    emb, cot = torch.randn(rows, net.input_ch, device=DEV), torch.randn(rows, 14, device=DEV)
    o64, g64 = grads64(emb, cot)
    res = {}
    for mode in ("dense", "library", "op_by_op"):
        dm.DENSE_TRUNK = mode == "dense"
        net.zero_grad(set_to_none=True)
        if mode == "op_by_op":
            h = net.trunk(emb)
            out = torch.cat([m(h) for _, m in net.heads()], -1)
        else:
            out = net.heads_from_embedding(emb)
        out.backward(cot)
        errs = {k: float((p.grad.double() - g64[k]).norm() / g64[k].norm()) for k, p in net.named_parameters()}
        res[mode] = {"out": float((out.double() - o64).norm() / o64.norm()), "worst": max(errs.values()), "first_layer": errs["linear.0.weight"], "last_layer": errs["linear.7.weight"]}
    print(json.dumps({"rows": rows, **res}))
