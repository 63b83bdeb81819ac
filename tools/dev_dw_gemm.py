"""Dev: how fast can dW = G^T X be formed for the node network's shapes? (R rows = time samples x nodes, 256 x 256 result)"""
import sys, torch
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 33280
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for out, inn in ((256, 256), (256, 340), (256, 84), (14, 256)):
    G, X = torch.randn(R, out, device=dev), torch.randn(R, inn, device=dev)
    ref = G.t().mm(X)
    res = {"mm(G.t, X)": bench(lambda: G.t().mm(X)), "mm(X.t, G).t": bench(lambda: X.t().mm(G).t())}
    Gt = G.t().contiguous()
    res["Gt contiguous + mm"] = bench(lambda: G.t().contiguous().mm(X))
    for g in (5, 13, 20, 26, 52, 65, 130, 260, 520, 1040, 2080):
        if R % g: continue
        f = lambda g=g: torch.bmm(G.view(g, R // g, out).transpose(1, 2), X.view(g, R // g, inn)).sum(0)
        err = float((f() - ref).abs().max() / ref.abs().max())
        res[f"bmm g={g} (K={R // g})"] = bench(f)
        assert err < 1e-4, err
        f2 = lambda g=g: torch.bmm(X.view(g, R // g, inn).transpose(1, 2), G.view(g, R // g, out)).sum(0).t()
        res[f"bmm' g={g}"] = bench(f2)
    flop = 2 * R * out * inn
    print(f"[{out} x {R}] x [{R} x {inn}]:", "  ".join(f"{k}: {v:.0f}us ({flop / v / 1e6:.0f} TF)" for k, v in sorted(res.items(), key=lambda kv: kv[1])[:6]), " | baseline %.0f" % res["mm(G.t, X)"])
