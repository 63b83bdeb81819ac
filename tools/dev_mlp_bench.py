#!/usr/bin/env python
"""dev: the fused deformation MLP alone (gsr_deform_mlp_forward / _backward) at config #3's batch: time per launch and the error against fp64,
with the fp32 library's error beside it. GSR_MLP_FP32=1 / GSR_MLP_RT=2 select the other kernels (read once per process)."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import deformation  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
in_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
dev = "cuda"
feat = torch.randn(n, in_dim, device=dev) * 0.5
shapes = [(64, in_dim), (64,)] + [s for o in (3, 3, 4) for s in ((64, 64), (64,), (o, 64), (o,))]
params = [torch.randn(*s, device=dev) * (0.12 if len(s) == 2 else 0.2) for s in shapes]
# parameter order of _FusedDeformMLP: W0, b0, then per head W1, b1, W2, b2
f = deformation._FusedDeformMLP
feat.requires_grad_(True)
ps = [p.clone().requires_grad_(True) for p in params]
def fwd():
    return f.apply(feat, *ps)
out = fwd()
cot = torch.randn_like(out)
def ref(dtype):
    F = feat.detach().to(dtype)[: 200_000]
    P = [p.detach().to(dtype) for p in params]
    a = torch.relu(F @ P[0].t() + P[1])
    outs = []
    for j in range(3):
        W1, b1, W2, b2 = P[2 + 4 * j: 6 + 4 * j]
        outs.append(torch.relu(a @ W1.t() + b1) @ W2.t() + b2)
    return torch.cat(outs, 1)
r64, r32 = ref(torch.float64), ref(torch.float32)
err = lambda a: float((a.double() - r64).abs().max() / r64.abs().max())
res = {"n": n, "in_dim": in_dim, "env": {k: os.environ.get(k) for k in ("GSR_MLP_FP32", "GSR_MLP_RT")},
       "fwd_err_vs_fp64": err(out[:200_000].detach()), "library_fp32_err_vs_fp64": err(r32)}
def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
with torch.no_grad():
    res["fwd_ms"] = timed(fwd)
def fb():
    for t in [feat] + ps: t.grad = None
    fwd().backward(cot)
res["fwd_bwd_ms"] = timed(fb, 5)
print(json.dumps(res))
