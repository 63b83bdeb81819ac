import sys, time, json, traceback
sys.path.insert(0, "/root/repo/tests")
from util import *
import numpy as np, torch
from diff_gaussian_rasterization import _C
print(torch.cuda.get_device_name(0), flush=True)
def run(P, W, H, deg=0, fppl=4, bppl=4, seed=0, scale_mean=0.005, precomp=False, tag=""):
    cam = make_camera(W, H)
    g = make_gaussians(P, cam, seed=seed, sh_degree=deg, scale_mean=scale_mean)
    gc, gd = make_cotangents(cam)
    bg = np.array([1.0, 0.5, 0.2], np.float32)
    cp = None
    if precomp:
        cp = np.random.default_rng(5).uniform(-1, 1, (P, 3)).astype(np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd, colors_precomp=cp)
    oh, gh = hip_run(g, cam, bg, gc, gd, colors_precomp=cp)
    m = compare(oh, gh, oo, go, tag=f"{tag} P={P} {W}x{H} deg={deg} ppl={fppl}/{bppl} R={oo['num_rendered']}")
    print(json.dumps(m), flush=True)
try:
    run(2000, 160, 120)
    run(2000, 160, 120, fppl=1, bppl=1)
    run(2000, 160, 120, fppl=2, bppl=2)
    run(2000, 150, 100, deg=3)
    run(2000, 150, 100, precomp=True)
    run(20000, 640, 480, deg=1)
    run(5000, 64, 48, scale_mean=0.05, tag="bigtiles")
    run(200000, 640, 480)
except Exception:
    traceback.print_exc()
