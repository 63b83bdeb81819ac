#!/usr/bin/env python
"""dev: where a tracking iteration's time goes -- eager (host waits for the instance count), eager in lazy mode (the host only enqueues),
hipGraph replay; for each the host's enqueue time per iteration and the wall time per iteration."""
import json, os, sys, time, types
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _C  # noqa: E402
from slam.camera import Camera, fov_from_focal, getProjectionMatrix2  # noqa: E402
from slam.system import default_config  # noqa: E402
from slam.tracking_graph import TrackingGraph  # noqa: E402
from synthetic_scene import GaussianModelStub, make_camera, make_gaussians  # noqa: E402

W, H = 640, 480
fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6
proj = getProjectionMatrix2(0.01, 100.0, cx, cy, fx, fy, W, H).transpose(0, 1)
cfg = default_config()
rng = np.random.default_rng(0)
rows = []
for P in [int(a) for a in sys.argv[1:] if a.isdigit()] or [10_000, 50_000]:
    g = make_gaussians(P, make_camera(W, H), seed=0, sh_degree=0)
    pc = GaussianModelStub(g, isotropic=False, dyn_frac=0.0, seed=0)
    pc.optimizer = None
    img = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device="cuda")
    depth = rng.uniform(0.5, 5, (H, W)).astype(np.float32)
    cam = Camera(1, img, depth, torch.eye(4), proj, fx, fy, cx, cy, fov_from_focal(fx, W), fov_from_focal(fy, H), H, W, 0.0)
    cam.compute_grad_mask(cfg)
    bg = torch.ones(3, device="cuda")
    pipe = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    tg = TrackingGraph(pc, pipe, bg, cfg, cam)
    tg.load(cam)
    n = 300
    def timed(fn):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return {"host_enqueue_us": (t1 - t0) / n * 1e6, "wall_us": (t2 - t0) / n * 1e6}
    row = {"gaussians": P}
    row["eager"] = timed(tg.iteration)
    _C.set_option("lazy", 1)
    row["eager_lazy"] = timed(tg.iteration)
    _C.set_option("lazy", 0)
    tg.load(cam); tg.capture(); tg.load(cam)
    row["graph"] = timed(tg.graph.replay)
    rows.append(row)
print(json.dumps(rows))
