import sys, os, cProfile, pstats, io
sys.path.insert(0, "/root/repo/4dgs-slam_amd")
from synthetic_scene import make_camera, make_gaussians, make_cotangents, keyframe_pose
import numpy as np, torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
cam = make_camera(640, 480); P = int(os.environ.get("P", 10000))
g = make_gaussians(P, cam, seed=0); gc, gd = make_cotangents(cam)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
rs = GaussianRasterizationSettings(480, 640, cam.tanfovx, cam.tanfovy, T([1,1,1]), 1.0, T(cam.viewmatrix), T(cam.projmatrix), T(cam.projmatrix_raw), 0, T(cam.campos), False, False)
m3, sh, op, sc, ro = T(g["means3D"], True), T(g["shs"], True), T(g["opacities"], True), T(g["scales"], True), T(g["rotations"], True)
th, rh = T(np.zeros(3), True), T(np.zeros(3), True)
m2 = torch.zeros_like(m3, requires_grad=True)
gcol, gdep = T(gc), T(gd)
rast = GaussianRasterizer(rs)
def step():
    for p in (m3, sh, op, sc, ro, th, rh, m2): p.grad = None
    c, r, d, o, n = rast(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=ro, theta=th, rho=rh)
    torch.autograd.backward([c, d], [gcol, gdep])
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])

# where does run_backward spend its time? time the pieces that run on the autograd thread
import time
from diff_gaussian_rasterization import _C, autograd as ag
acc = {"glue_bwd": 0.0, "py_bwd": 0.0, "n": 0}
orig_glue = _C._glue.rasterize_gaussians_backward_fused
def timed_glue(*a):
    t0 = time.perf_counter(); r = orig_glue(*a); acc["glue_bwd"] += time.perf_counter() - t0; return r
_C._glue.rasterize_gaussians_backward_fused = timed_glue
orig_bwd = ag._RasterizeGaussians.backward
def timed_bwd(ctx, *g):
    t0 = time.perf_counter(); r = orig_bwd(ctx, *g); acc["py_bwd"] += time.perf_counter() - t0; acc["n"] += 1; return r
ag._RasterizeGaussians.backward = staticmethod(timed_bwd)
for _ in range(20): step()
torch.cuda.synchronize()
acc.update(glue_bwd=0.0, py_bwd=0.0, n=0)
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 300 * 1e6
n_ = max(acc["n"], 1)      # (0 with the C++ autograd node: the Python Function is not on the path)
print("step %.1f us; Function.backward %.1f us of which glue call %.1f us" % (tot, acc["py_bwd"] / n_ * 1e6, acc["glue_bwd"] / n_ * 1e6))
