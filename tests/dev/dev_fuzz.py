"""Randomised parity sweep (HIP vs oracle) over scene shapes: image sizes, Gaussian counts, scales (list lengths from a few to
several thousand per tile, crossing the 1024 / 4096 sort thresholds), SH degrees; consecutive scenes in one process so that the
speculative binning capacity is alternately too small and too large."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/4dgs-slam_amd")
from util import oracle_run, hip_run, compare, make_camera, make_gaussians, make_cotangents
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    W, H = int(rng.integers(17, 330)), int(rng.integers(17, 250))
    P = int(rng.choice([1, 7, 300, 3000, 12000, 30000]))
    sm = float(rng.choice([0.002, 0.01, 0.05, 0.2]))
    deg = int(rng.integers(0, 4))
    cam = make_camera(W, H)
    g = make_gaussians(P, cam, seed=int(rng.integers(1 << 30)), sh_degree=deg, scale_mean=sm)
    gc, gd = make_cotangents(cam, seed=it)
    bg = rng.uniform(0, 1, 3).astype(np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    oh, gh = hip_run(g, cam, bg, gc, gd)
    m = compare(oh, gh, oo, go)
    R = int(st.state()["R"]) if "R" in st.state() else -1
    worst = max(v for k, v in m.items() if isinstance(v, float))
    flag = worst > 1e-3 or m["radii_mismatch"] or m["color"] > 1e-4
    bad += bool(flag)
    print(("BAD " if flag else "ok  ") + "W%3d H%3d P%5d sm%.3f deg%d  worst %.2e color %.1e radii_mm %d nt_mm %d" % (W, H, P, sm, deg, worst, m["color"], m["radii_mismatch"], m["n_touched_mismatch"]), flush=True)
print("bad cases:", bad)
