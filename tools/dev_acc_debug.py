import sys, numpy as np, torch
sys.path[:0] = ["/root/repo/tests", "/root/repo", "/root/repo/4dgs-slam_amd"]
from util import make_camera, make_gaussians, make_cotangents, keyframe_pose
from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
from mapping_shard import GradBucket
def _settings(cam, dev="cuda"):
    T = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    return GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, T([1, 1, 1]), 1.0, T(cam.viewmatrix), T(cam.projmatrix),
                                         T(cam.projmatrix_raw), 0, T(cam.campos), False, False)
P = 120_000
g = make_gaussians(P, make_camera(640, 480), seed=3)
T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
def run(fused, views):
    params = [T(g[k], True) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    bucket = GradBucket(params).attach(fused_accumulate=fused)
    bucket.zero_grads()
    for k in views:
        R, t = keyframe_pose(k)
        cam = make_camera(640, 480, R=R, t=t)
        gc, gd = make_cotangents(cam, seed=k)
        means2D = torch.zeros_like(params[0], requires_grad=True)
        color, radii, depth, opacity, n_touched = GaussianRasterizer(_settings(cam))(
            means3D=params[0], means2D=means2D, opacities=params[2], shs=params[1], scales=params[3], rotations=params[4])
        ((color * T(gc)).sum() + (depth * T(gd)).sum()).backward()
    return [p.grad.clone() for p in params]
for views in ((0,), (0, 3), (0, 3, 6)):
    a, b = run(True, views), run(False, views)
    for n, x, y in zip(("means3D", "shs", "opacity", "scales", "rot"), a, b):
        d = (x - y).abs()
        print(views, n, "max diff %.3e" % float(d.max()), "n diff", int((d > 0).sum()), "nan", int(torch.isnan(x).sum()), int(torch.isnan(y).sum()),
              "rel %.2e" % float(d.max() / y.abs().max()))
