#!/usr/bin/env python
"""optimizer.step() over the reference's six parameter groups (200k Gaussians, SH degree 0): torch.optim.Adam vs FusedAdam."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "4dgs-slam_amd"))
from fused_adam import FusedAdam
P = 200_000
shapes = [("xyz", (P, 3)), ("f_dc", (P, 1, 3)), ("f_rest", (P, 0, 3)), ("opacity", (P, 1)), ("scaling", (P, 3)), ("rotation", (P, 4))]
out = {}
for name, cls in (("torch_adam", torch.optim.Adam), ("fused_adam", FusedAdam)):
    groups = [{"params": [torch.nn.Parameter(torch.randn(*s, device="cuda"))], "lr": 1e-3, "name": n} for n, s in shapes if 0 not in s]
    opt = cls(groups, lr=0.0, eps=1e-15)
    for g in groups:
        g["params"][0].grad = torch.randn_like(g["params"][0])
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        opt.step()
    torch.cuda.synchronize()
    out[name + "_us_per_step"] = (time.perf_counter() - t0) / 200 * 1e6
print(json.dumps(out))
