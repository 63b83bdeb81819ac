"""dev: the node network's forward / input-gradient GEMM shapes under torch's BLAS back-ends."""
import torch
dev = "cuda:0"
def bench(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for lib in ("default", "hipblaslt", "rocblas", "ck"):
    try:
        if lib != "default":
            torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable:", e); continue
    for R in (33280, 69632):
        X, W, b = torch.randn(R, 256, device=dev), torch.randn(256, 256, device=dev), torch.randn(256, device=dev)
        G = torch.randn(R, 256, device=dev)
        try:
            t1 = bench(lambda: torch._addmm_activation(b, X, W.t(), use_gelu=False))
            t2 = bench(lambda: G.mm(W))
            t3 = bench(lambda: torch.addmm(b, X, W.t()))
            fl = 2 * R * 256 * 256
            print(lib, R, "addmm_act %.1f us (%.0f TF)  mm dX %.1f us (%.0f TF)  addmm %.1f us" % (t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3))
        except Exception as e:
            print(lib, R, "failed:", str(e)[:100])
