import sys, time, os
sys.path.insert(0, "/root/repo/4dgs-slam_amd")
from synthetic_scene import make_camera, make_gaussians, make_cotangents, keyframe_pose
import numpy as np, torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
cam = make_camera(640, 480); P = int(os.environ.get('P', 200000))
g = make_gaussians(P, cam, seed=0); gc, gd = make_cotangents(cam)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
rs = GaussianRasterizationSettings(480, 640, cam.tanfovx, cam.tanfovy, T([1,1,1]), 1.0, T(cam.viewmatrix), T(cam.projmatrix), T(cam.projmatrix_raw), 0, T(cam.campos), False, False)
m3, sh, op, sc, ro = T(g["means3D"], True), T(g["shs"], True), T(g["opacities"], True), T(g["scales"], True), T(g["rotations"], True)
gcol, gdep = T(gc), T(gd)
rast = GaussianRasterizer(rs)
def step(sync_mid=False):
    for p in (m3, sh, op, sc, ro): p.grad = None
    m2 = torch.zeros_like(m3, requires_grad=True)
    t0 = time.perf_counter()
    c, r, d, o, n = rast(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=ro)
    t1 = time.perf_counter()
    if sync_mid: torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.autograd.backward([c, d], [gcol, gdep])
    t3 = time.perf_counter()
    return t1 - t0, t3 - t2
for _ in range(10): step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
print("mailbox=%s pipelined: %.1f us/step" % (os.environ.get("GSR_MAILBOX", "1"), (time.perf_counter() - t0) / N * 1e6))
f = b = 0
for _ in range(N):
    torch.cuda.synchronize()
    a, c_ = step(sync_mid=True); f += a; b += c_
    torch.cuda.synchronize()
print("host time of forward call (incl. wait for R): %.1f us; host time of backward call (launch only): %.1f us" % (f / N * 1e6, b / N * 1e6))
