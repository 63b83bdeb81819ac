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
