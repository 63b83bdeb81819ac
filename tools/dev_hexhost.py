import os, sys, time, cProfile, pstats
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
import hexplane
dev = "cuda"
field = hexplane.HexPlaneField(1.6, {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}, [1, 2, 4, 8]).to(dev)
n = 8192
pts = torch.rand(n, 3, device=dev, requires_grad=True)
tim = torch.full((n, 1), 0.3, device=dev)
cot = torch.randn(n, 128, device=dev)
def fb():
    for lv in field.grids:
        for p in lv:
            p.grad = None
    pts.grad = None
    (field(pts, tim) * cot).sum().backward()
for _ in range(5): fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): fb()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per iter %.0f us, +sync tail %.0f us" % ((t1 - t0) / 50 * 1e6, (t2 - t1) * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): fb()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
