"""Mapping-iteration rasterizer work at SLAM scale, view by view vs through the multi-view entry point:
P Gaussians x V keyframes @640x480, forward + backward of every view, gradients accumulated in an attached bucket.
    python tools/bench_views.py [--gaussians 30000] [--views 10] [--scale-mean 0.03] [--iters 50]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")]
import torch
from test_hip_views import _scene, _single, _multi
from mapping_shard import GradBucket
from diff_gaussian_rasterization import _C

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=30000)
ap.add_argument("--views", type=int, default=10)
ap.add_argument("--scale-mean", type=float, default=0.03)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--dyn", action="store_true")
args = ap.parse_args()
par, settings, cots, slot, deltas, poses = _scene(P=args.gaussians, V=args.views, W=640, H=480, dyn=args.dyn, scale_mean=args.scale_mean)
plist = [p for p in (par["xyz"], par["f_dc"], par["f_rest"], par["logit"], par["log_scales"], par["rot"]) if p.numel()]
bucket = GradBucket(plist).attach()


def run(fn, n):
    for _ in range(5):
        bucket.zero_grads(); fn(par, settings, cots, slot, deltas, poses)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bucket.zero_grads(); fn(par, settings, cots, slot, deltas, poses)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def kernels(fn):
    _C.profile_reset(); _C.profile_enable(True)
    for _ in range(5):
        bucket.zero_grads(); fn(par, settings, cots, slot, deltas, poses)
    torch.cuda.synchronize(); _C.profile_enable(False)
    return {k: round(v[0] / 5 * 1e3, 1) for k, v in _C.profile_read().items() if v[1]}     # us per iteration (all views)


one = run(_single, args.iters)
many = run(_multi, args.iters)
out = {"workload": f"{args.gaussians} Gaussians (scale_mean {args.scale_mean}) x {args.views} views @640x480, fwd+bwd per view, accumulated gradients" + (", control-node deltas on 25 %" if args.dyn else ""),
       "ms_per_iteration_view_by_view": one, "ms_per_iteration_multi_view": many, "speedup": one / many,
       "batched_calls": _C.set_option("views_batched"), "kernel_us_per_iteration_view_by_view": kernels(_single), "kernel_us_per_iteration_multi_view": kernels(_multi)}
print(json.dumps(out))
