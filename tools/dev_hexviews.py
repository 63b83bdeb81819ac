#!/usr/bin/env python
"""Development timing of the batched-views HexPlane field (gsr_hexplane_*_views) at config #3's size, kernel by kernel (torch profiler-free:
HIP events around the autograd calls).  usage: python tools/dev_hexviews.py [--P 500000] [--views 8]"""
import argparse, json, os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import deformation, hexplane
from synthetic_scene import make_camera, make_gaussians
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=500_000)
ap.add_argument("--views", type=int, default=8)
a = ap.parse_args()
g = make_gaussians(a.P, make_camera(640, 480), seed=0, sh_degree=0)
xyz = torch.tensor(g["means3D"], device="cuda", requires_grad=True)
net = deformation.deform_network(deformation.default_hidden_params(bounds=8.0), "cuda").to("cuda")
field = net.deformation_net.grid
times = [k / max(a.views - 1, 1) * 2 - 1 for k in range(a.views)]
cot = torch.randn((a.views, a.P, field.feat_dim), device="cuda")
cot[:, torch.rand(a.P, device="cuda") < 0.08] = 0
def run():
    f = field.forward_views(xyz, times)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f.backward(cot)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(2):
    run()
print(json.dumps({"P": a.P, "views": a.views, "dbg": os.environ.get("GSR_HEXV_DEBUG", "0"), "backward_ms": round(min(run() for _ in range(4)), 3)}))
