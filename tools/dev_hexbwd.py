#!/usr/bin/env python
"""dev: where does the HexPlane backward spend its time? (time planes vs spatial planes, uniform vs random t)"""
import os, sys, types
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import hexplane
from tools.bench_deformation import timeit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = "cuda"
field = hexplane.HexPlaneField(1.6, {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}, [1, 2, 4, 8]).to(dev)
rng = np.random.default_rng(0)
pts = torch.tensor(rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32), device=dev)
cot = torch.randn(n, 128, device=dev)
for tname, tim in (("uniform t", torch.full((n, 1), 0.3, device=dev)), ("random t", torch.rand(n, 1, device=dev) * 2 - 1)):
    for which, sel in ((("all", lambda p: True),) if os.environ.get("GSR_HEX_XCD_STRIDE") else (("all", lambda p: True), ("spatial only", lambda p: p in (0, 1, 3)), ("time only", lambda p: p in (2, 4, 5)), ("none (xyz only)", lambda p: False))):
        for lv in field.grids:
            for p, plane in enumerate(lv):
                plane.requires_grad_(sel(p))
        x = pts.clone().requires_grad_(True)
        def fb():
            for lv in field.grids:
                for plane in lv:
                    plane.grad = None
            x.grad = None
            (field(x, tim) * cot).sum().backward()
        print("%-10s %-16s fwd+bwd %8.1f us" % (tname, which, timeit(fb, 10)))
