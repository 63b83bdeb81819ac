import sys, json
sys.path.insert(0, "/root/repo/tests")
from util import *
import numpy as np, torch
from simple_knn._C import distCUDA2
from scipy.spatial import cKDTree
def check(pts, tag):
    pts = np.ascontiguousarray(pts, np.float32)
    ref = oracle.knn_dist2(pts) if len(pts) <= 20000 else None
    out = distCUDA2(torch.tensor(pts, device="cuda")).cpu().numpy()
    if len(pts) >= 4:
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        kd = (d[:, 1:] ** 2).mean(1)
        print(tag, len(pts), "vs kdtree max rel", float(np.max(np.abs(out - kd) / np.maximum(kd, 1e-30))), end=" ")
    if ref is not None:
        fin = np.isfinite(ref)
        print("vs oracle max rel", float(np.max(np.abs(out[fin] - ref[fin]) / np.maximum(ref[fin], 1e-30))) if fin.any() else None, "inf match", bool((np.isinf(out) == np.isinf(ref)).all()))
    else:
        print()
rng = np.random.default_rng(0)
check(rng.uniform(-1, 1, (2400, 3)), "uniform")
check(rng.uniform(-1, 1, (9600, 3)) * [5, 1, 0.01], "flat")
check(np.concatenate([rng.normal(0, 0.01, (3000, 3)), rng.normal(5, 2, (3000, 3))]), "clustered")
check(rng.uniform(-1, 1, (3, 3)), "tiny3")
check(rng.uniform(-1, 1, (1, 3)), "tiny1")
check(np.repeat(rng.uniform(-1, 1, (100, 3)), 5, 0), "dupes")
check(np.stack([np.linspace(0, 1, 1000), np.zeros(1000), np.zeros(1000)], 1), "line")
check(rng.uniform(-3, 3, (300000, 3)), "big")
import time
p = torch.tensor(rng.uniform(-3, 3, (9600, 3)).astype(np.float32), device="cuda")
for _ in range(3): distCUDA2(p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): distCUDA2(p)
torch.cuda.synchronize(); print("knn 9600 pts: %.1f us/call" % ((time.perf_counter() - t0) / 20 * 1e6))
