#!/usr/bin/env python
"""Development: eight chained dense layers (+ a tiny kernel between them) as a hipGraph replay -- what a launch costs beside its own duration."""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import dense_layers as dl
R = int(sys.argv[1]) if len(sys.argv) > 1 else 33280
between = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda"
N = K = 256
X = torch.randn((R, K), device=dev)
Ws = [torch.randn((N, K), device=dev) / 16 for _ in range(8)]
b = torch.randn((N,), device=dev)
planes = [dl.split_weight(w) for w in Ws]
bufs = [torch.empty((R, N), device=dev) for _ in range(8)]
small = torch.zeros((1024,), device=dev)
def chain():
    h = X
    for i in range(8):
        h = dl.dense_forward(h, planes[i], N, K, b, relu=True, out=bufs[i])
        for _ in range(between):
            small.add_(1.0)
def lib_chain():
    h = X
    for i in range(8):
        h = torch._addmm_activation(b, h, Ws[i].t(), use_gelu=False)
        for _ in range(between):
            small.add_(1.0)
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
def replayed(f, n=20):
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
print(json.dumps({"rows": R, "between": between, "wide": os.environ.get("GSR_DENSE_WIDE", "1"), "chain_replayed_us": round(replayed(chain), 1),
                  "chain_eager_us": round(timed(chain), 1), "library_replayed_us": round(replayed(lib_chain), 1)}))
