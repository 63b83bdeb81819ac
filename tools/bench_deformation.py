#!/usr/bin/env python
"""Timing of the deformation field at the shipped geometry (HexPlane [64,64,64,25] x multires [1,2,4,8],
32 features per plane, MLP width 64): the fused HIP field vs the reference's tensor program (24 F.grid_sample calls on [C][H][W]
planes + products + concat) on the same GPU, forward and forward+backward, and the whole deform_network around each.
Prints one JSON line.  usage: python tools/bench_deformation.py [--n 200000] [--iters 20]"""
import argparse
import itertools
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import deformation  # noqa: E402
import hexplane     # noqa: E402


def torch_field(pts, tim, aabb, levels):
    p = torch.clamp((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, -1.0, 1.0)
    p4 = torch.cat((p, tim), dim=-1)
    feats = []
    for planes in levels:
        prod = 1.0
        for (c0, c1), plane in zip(itertools.combinations(range(4), 2), planes):
            s = F.grid_sample(plane, p4[:, [c0, c1]].view(1, 1, -1, 2), align_corners=True, mode="bilinear", padding_mode="border")
            prod = prod * s.view(plane.shape[1], -1).t()
        feats.append(prod)
    return torch.cat(feats, dim=-1)


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only-network", action="store_true", help="time only the whole network fwd+bwd (for profiling)")
    ap.add_argument("--visible", type=float, default=1.0, help="fraction of points with a non-zero cotangent (the rest: Gaussians the view does not see)")
    ap.add_argument("--sorted", action="store_true", help="points in Morton-like (cell) order instead of random order")
    a = ap.parse_args()
    dev = "cuda"
    args = deformation.default_hidden_params()
    torch.manual_seed(0)
    net = deformation.deform_network(args, dev).to(dev)
    field = net.deformation_net.grid
    rng = np.random.default_rng(0)
    pts_np = rng.uniform(-1.5, 1.5, size=(a.n, 3)).astype(np.float32)
    if a.sorted:
        cell = np.floor((pts_np + 1.6) / 3.2 * 512).astype(np.int64)            # Morton order of the finest xy cells
        def spread(v):
            v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
            return v
        pts_np = pts_np[np.argsort(spread(cell[:, 0]) | (spread(cell[:, 1]) << 1), kind="stable")]
    pts = torch.tensor(pts_np, device=dev, requires_grad=True)
    tim = torch.full((a.n, 1), 0.3, device=dev)
    scales = torch.randn(a.n, 3, device=dev, requires_grad=True)
    rots = torch.randn(a.n, 4, device=dev, requires_grad=True)
    opac = torch.randn(a.n, 1, device=dev)
    shs = torch.randn(a.n, 16, 3, device=dev)
    cot = torch.randn(a.n, 128, device=dev)
    vis = (torch.rand(a.n, 1, device=dev) < a.visible).float()
    cot = cot * vis
    ref_levels = [[p.detach().clone().contiguous().requires_grad_(True) for p in lv] for lv in field.grids]

    def zero():
        for p in net.parameters():
            p.grad = None
        for lv in ref_levels:
            for p in lv:
                p.grad = None
        pts.grad = None

    res = {"n": a.n, "sorted": a.sorted, "unit": "us"}
    if a.only_network:
        def net_fb0():
            zero()
            o = net(pts, scales, rots, opac, shs, tim)
            ((o[3] * vis).sum() + (o[4] * vis).sum() + (o[5] * vis).sum()).backward()
        res["network_fwdbwd_fused_field"] = round(timeit(net_fb0, a.iters), 1)
        res["visible"] = a.visible
        print(json.dumps(res))
        return
    with torch.no_grad():
        res["field_fwd_fused"] = timeit(lambda: field(pts, tim), a.iters)
        res["field_fwd_torch"] = timeit(lambda: torch_field(pts, tim, field.aabb, ref_levels), a.iters)

    def fb_fused():
        zero()
        (field(pts, tim) * cot).sum().backward()

    def fb_torch():
        zero()
        (torch_field(pts, tim, field.aabb, ref_levels) * cot).sum().backward()

    res["field_fwdbwd_fused"] = timeit(fb_fused, a.iters)
    res["field_fwdbwd_torch"] = timeit(fb_torch, max(3, a.iters // 4))

    def net_fb():
        zero()
        o = net(pts, scales, rots, opac, shs, tim)
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()

    res["network_fwdbwd_fused_field"] = timeit(net_fb, a.iters)
    with torch.no_grad():
        res["network_fwd_fused_field"] = timeit(lambda: net(pts, scales, rots, opac, shs, tim), a.iters)
    # the same network with the reference's field program in place of the fused one
    orig = hexplane.hexplane_features
    hexplane.hexplane_features = lambda p, t, aabb, grids: torch_field(p, t, aabb, ref_levels)
    try:
        res["network_fwdbwd_torch_field"] = timeit(net_fb, max(3, a.iters // 4))
    finally:
        hexplane.hexplane_features = orig
    res = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}
    res["speedup_field_fwd"] = round(res["field_fwd_torch"] / res["field_fwd_fused"], 2)
    res["speedup_field_fwdbwd"] = round(res["field_fwdbwd_torch"] / res["field_fwdbwd_fused"], 2)
    res["speedup_network_fwdbwd"] = round(res["network_fwdbwd_torch_field"] / res["network_fwdbwd_fused_field"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
