#!/usr/bin/env python
"""Timing of the SC-GS control-node warp at the shipped sizes (512 nodes, K = 3, local frame, residual
rotation): the fused HIP op vs the reference's tensor program (utils/time_utils.py:981-1011,1199-1258) with a brute-force
cdist + topk standing in for pytorch3d.ops.knn_points (which has no ROCm build), forward and forward+backward.
Prints one JSON line.  usage: python tools/bench_control_nodes.py [--n 100000] [--m 512] [--iters 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import control_nodes as cn  # noqa: E402
from tools.bench_deformation import timeit  # noqa: E402


def torch_program(x, mask, nodes, rr, wr, tr, ro, sc, lr, K):
    """the reference's statement, on the GPU"""
    d = ((x[:, None, :] - nodes[None, :, :3].detach()) ** 2).sum(-1)
    nn_dist, idx = torch.topk(d, K, dim=-1, largest=False, sorted=True)
    w = torch.exp(-nn_dist / (2 * torch.exp(rr)[idx] ** 2)) * torch.sigmoid(wr)[idx][..., 0] + 1e-7
    w = w / w.sum(dim=-1, keepdim=True)
    bias = torch.tensor([1.0, 0, 0, 0], device=x.device)
    R = cn.quaternion_to_matrix(lr + bias)
    nn_nodes = nodes[idx][..., :3].detach()
    Ax = torch.einsum("nkab,nkb->nka", R[idx], x[:, None] - nn_nodes) + nn_nodes + tr[idx]
    translate = ((Ax * w[..., None]).sum(dim=1) - x) * mask
    rotation = (ro[idx] * w[..., None]).sum(dim=1) * mask
    scale = (sc[idx] * w[..., None]).sum(dim=1) * mask
    return translate, rotation, scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    rng = np.random.default_rng(0)
    T = lambda arr, rg=False: torch.tensor(np.asarray(arr, np.float32), device=dev, requires_grad=rg)
    x, mask = T(rng.uniform(-1, 1, size=(a.n, 3))), torch.ones(a.n, 1, device=dev)
    nodes = T(rng.uniform(-1, 1, size=(a.m, 3)))
    rr, wr = T(np.log(rng.uniform(0.1, 0.4, size=a.m)), True), T(rng.normal(size=(a.m, 1)), True)
    tr, ro = T(rng.normal(scale=0.1, size=(a.m, 3)), True), T(rng.normal(scale=0.1, size=(a.m, 4)), True)
    sc, lr = T(rng.normal(scale=0.1, size=(a.m, 3)), True), T(rng.normal(scale=0.2, size=(a.m, 4)), True)
    leaves = (rr, wr, tr, ro, sc, lr)

    def fused():
        r = cn.node_blend(x, mask, nodes, rr, wr, tr, ro, sc, lr, K=3)
        return r["d_xyz"], r["d_rotation"], r["d_scaling"]

    def fb(fn):
        def run():
            for t in leaves:
                t.grad = None
            o = fn()
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()
        return run

    res = {"n": a.n, "nodes": a.m, "K": 3, "unit": "us"}
    with torch.no_grad():
        res["fwd_fused"] = timeit(fused, a.iters)
        res["fwd_torch"] = timeit(lambda: torch_program(x, mask, nodes, rr, wr, tr, ro, sc, lr, 3), a.iters)
        p1, p2 = x[None], nodes[None]
        res["knn_points_fused"] = timeit(lambda: cn.knn_points(p1, p2, K=3), a.iters)
        res["knn_points_torch_topk"] = timeit(lambda: torch.topk(((x[:, None] - nodes[None]) ** 2).sum(-1), 3, largest=False), a.iters)
    res["fwdbwd_fused"] = timeit(fb(fused), a.iters)
    res["fwdbwd_torch"] = timeit(fb(lambda: torch_program(x, mask, nodes, rr, wr, tr, ro, sc, lr, 3)), a.iters)
    res = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}
    res["speedup_fwd"] = round(res["fwd_torch"] / res["fwd_fused"], 2)
    res["speedup_fwdbwd"] = round(res["fwdbwd_torch"] / res["fwdbwd_fused"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
