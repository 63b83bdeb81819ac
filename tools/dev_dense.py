#!/usr/bin/env python
"""Development timing of the bf16x3 dense layers (include/dense_layers.h) against the library's fp32 GEMMs at the node network's shapes.
usage: python tools/dev_dense.py [rows]"""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import dense_layers as dl
from slam.deform_model import _grad_weight
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
dev = "cuda"
def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = {"rows": R}
for (N, K) in ((256, 84), (256, 256), (256, 340)):
    X, W, b, G = torch.randn((R, K), device=dev), torch.randn((N, K), device=dev) / 16, torch.randn((N,), device=dev), torch.randn((R, N), device=dev)
    planes, planes_t = dl.split_weight(W), dl.split_weight(W, transposed=True)
    Y = torch.empty((R, N), device=dev)
    out[f"{N}x{K}"] = {
        "split_us": round(timed(lambda: dl.split_weight(W, out=planes)), 1),
        "forward_us": round(timed(lambda: dl.dense_forward(X, planes, N, K, b, relu=True, out=Y)), 1),
        "forward_lib_fp32_us": round(timed(lambda: torch._addmm_activation(b, X, W.t(), use_gelu=False)), 1),
        "dgrad_us": round(timed(lambda: dl.dense_forward(G, planes_t, K, N, gate=Y)), 1),
        "dgrad_lib_fp32_us": round(timed(lambda: G.mm(W)), 1),
        "wgrad_us": round(timed(lambda: dl.dense_wgrad(G, X)), 1),
        "wgrad_lib_fp32_us": round(timed(lambda: G.t().mm(X)), 1),
        "wgrad_lib_row_groups_us": round(timed(lambda: _grad_weight(G, X)), 1),
        "tflops_fp32_equivalent_forward": None}
    e = out[f"{N}x{K}"]
    e["tflops_fp32_equivalent_forward"] = round(2.0 * R * N * K / e["forward_us"] / 1e6, 1)
print(json.dumps(out))
# ---- the whole trunk forward: one layer-fused launch (+ ten weight splits) against the library's eight GEMMs with ReLU epilogue + heads ----
import numpy as np
E, W, D, skip = 84, 256, 8, 4
emb = torch.randn((R, E), device=dev)
Ws = [torch.randn((W, E if k == 0 else (E + W if k == skip + 1 else W)), device=dev) * 0.06 for k in range(D)]
bs = [torch.randn((W,), device=dev) * 0.1 for _ in range(D)]
Wh, bh = torch.randn((14, W), device=dev) * 0.05, torch.randn((14,), device=dev)
def lib_forward():
    h = emb
    for k in range(D):
        h = torch._addmm_activation(bs[k], h, Ws[k].t(), use_gelu=False)
        if k == skip:
            h = torch.cat([emb, h], -1)
    return torch.addmm(bh, h, Wh.t())
def replayed(f, n=20):
    """f captured as a hipGraph and replayed: device time without the host's launch overhead (what the SLAM loops' replays see)"""
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    return timed(g.replay, n)
print(json.dumps({"rows": R, "trunk_forward_fused_us": round(timed(lambda: dl.trunk_forward(emb, Ws, bs, Wh, bh), 10), 1),
                  "trunk_forward_library_us": round(timed(lib_forward, 10), 1),
                  "replayed_trunk_forward_fused_us": round(replayed(lambda: dl.trunk_forward(emb, Ws, bs, Wh, bh)), 1),
                  "replayed_trunk_forward_library_us": round(replayed(lib_forward), 1)}))
