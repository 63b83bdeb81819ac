#!/usr/bin/env python
"""bench.py -- rasterized Gaussians/s, forward+backward, 200k Gaussians @640x480 (BASELINE.json metric, configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input: GaussianRasterizer forward + autograd backward
(colour + depth cotangents) of P=200 000 Gaussians on one 640x480 view, inputs resident in HBM before the timed
region. With N>1 every rank renders its own keyframe of the same Gaussian set (weak scaling, view sharding of
SURVEY.md 8e) and the flattened Gaussian gradient block is summed with ONE RCCL all-reduce per step -- the step of the
mapping back-end that north_star shards. value = N * P * K / max-over-ranks(time).

One JSON line on rank 0, with `roofline` (dominant kernel = render_bwd, timed with HIP events on its launch stream
inside the timed region) and `cpu_baseline` (the C oracle = a sequential port of the reference algorithm, timed on
this box's host CPU, 1 core). The oracle is imported ONLY for that leg.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "4dgs-slam_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

P_GAUSS, WIDTH, HEIGHT = 200_000, 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(P, V, R, N, M):
    """minimum traffic of the reference algorithm, each item moved once, whole fwd+bwd and the
    render-backward kernel's share (per instance: 44 B read + 40 B accumulated; per pixel: 24 B read)."""
    total = 60 * P + (290 + 36 * M) * V + 292 * R + 52 * N
    render_bwd = 84 * R + 24 * N
    return total, render_bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=P_GAUSS)
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # test hooks (tests/test_hip_configs.py runs the N > 1 code path on a 1-GPU box): GSR_BENCH_DEVICE pins every rank to one
    # device, GSR_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on the same GPU
    local_dev = int(os.environ.get("GSR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("GSR_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    from mapping_shard import GradBucket
    from synthetic_scene import make_camera, make_gaussians, make_cotangents, keyframe_pose

    P = args.gaussians
    R_w, t_w = keyframe_pose(rank)  # rank r renders keyframe r of the same scene
    cam = make_camera(WIDTH, HEIGHT, R=R_w, t=t_w)
    g = make_gaussians(P, make_camera(WIDTH, HEIGHT), seed=0, sh_degree=args.sh_degree)
    gc, gd = make_cotangents(cam, seed=1 + rank)
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)
    rs = GaussianRasterizationSettings(
        image_height=HEIGHT, image_width=WIDTH, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=T([1.0, 1.0, 1.0]),
        scale_modifier=1.0, viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw),
        sh_degree=args.sh_degree, campos=T(cam.campos), prefiltered=False, debug=False)
    rast = GaussianRasterizer(rs)
    means3D, shs, opac = T(g["means3D"], True), T(g["shs"], True), T(g["opacities"], True)
    scales, rots = T(g["scales"], True), T(g["rotations"], True)
    theta, rho = T(np.zeros(3), True), T(np.zeros(3), True)
    gcol, gdep = T(gc), T(gd)
    params = [means3D, shs, opac, scales, rots]   # same order as the rasterizer's gradient block
    bucket = GradBucket(params) if world > 1 else None
    stats = {}

    # screen-space gradient holder of the reference's render() (gaussian_renderer/__init__.py:71): an input of the rasterizer,
    # created once here -- filling it is the caller's work, not part of the rasterizer's forward + backward
    means2D = torch.zeros_like(means3D, requires_grad=True)

    def step():
        for p_ in params + [theta, rho, means2D]:
            p_.grad = None
        color, radii, depth, opacity, n_touched = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                                       scales=scales, rotations=rots, theta=theta, rho=rho)
        torch.autograd.backward([color, depth], [gcol, gdep])
        if bucket is not None:
            stats["allreduce"] = bucket.all_reduce_grads()   # in place on the backward's own output range when possible
        stats["radii"] = radii

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _C.profile_reset()
    _C.profile_enable(["render_bwd"])       # 2 event records per step on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    _C.profile_enable(False)
    dom_ms, dom_calls = _C.profile_read()["render_bwd"]
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel breakdown in a separate, untimed pass (all kernels timed)
    _C.profile_reset()
    _C.profile_enable(True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    _C.profile_enable(False)
    kern = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in _C.profile_read().items() if v[1]}  # us per launch

    if rank == 0:
        # workload facts from the run itself
        with torch.no_grad():
            nr, color, radii, *_ = _C.rasterize_gaussians(rs.bg, means3D, torch.Tensor([]), opac, scales, rots, 1.0, torch.Tensor([]),
                                                          rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, rs.tanfovx, rs.tanfovy,
                                                          HEIGHT, WIDTH, shs, args.sh_degree, rs.campos, False, False)
        V = int((radii > 0).sum().item())
        M = int(shs.shape[1])
        N = WIDTH * HEIGHT
        b_total, b_dom = algorithmic_bytes(P, V, nr, N, M)
        dom_s = dom_ms / max(dom_calls, 1) * 1e-3
        achieved = b_dom / dom_s / 1e9 if dom_s > 0 else 0.0
        traffic = None
        tfile = os.path.join(REPO, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tfile) and P == P_GAUSS:
            try:
                traffic = json.load(open(tfile)).get("render_bwd_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rasterized Gaussians/s fwd+bwd @640x480 (200k G)",
            "value": world * P * args.steps / dt,
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} static Gaussians, 1 cam @{WIDTH}x{HEIGHT} per GPU, SH degree {args.sh_degree}, fwd+bwd"
                                   + (f", {world} views sharded + RCCL all-reduce of {bucket.nbytes} B grads" if world > 1 else ""),
                       "visible": V, "instances": nr, "pixels": N, "host_binding": _C.binding(),
                       **({"allreduce": stats.get("allreduce")} if world > 1 else {})},
            "roofline": {"bound": "hbm", "kernel": "render_bwd", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_dom, "kernel_us": dom_s * 1e6,
                         "whole_step_algorithmic_bytes": b_total,
                         "whole_step_GBps": b_total / (dt / args.steps) / 1e9,
                         "pair_evals_per_s_bwd": nr * 256 / dom_s if dom_s > 0 else None},
            "kernel_us": kern,
        }
        if not args.no_cpu_baseline:
            import oracle  # CPU baseline leg only

            t1 = time.perf_counter()
            o, st = oracle.rasterize_forward(bg=np.ones(3, np.float32), means3D=g["means3D"], opacities=g["opacities"], shs=g["shs"],
                                             scales=g["scales"], rotations=g["rotations"], viewmatrix=cam.viewmatrix,
                                             projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                             image_height=HEIGHT, image_width=WIDTH, sh_degree=args.sh_degree)
            oracle.rasterize_backward(st, projmatrix_raw=cam.projmatrix_raw, dL_dcolor=gc, dL_ddepth=gd)
            t_cpu = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": P / t_cpu, "unit": "Gaussians/s", "cores": 1, "kind": "port",
                                   "sample": f"1 fwd+bwd of the full {P} Gaussians @{WIDTH}x{HEIGHT} workload ({t_cpu:.1f} s), "
                                             f"oracle/gs_oracle.c single-threaded on {os.cpu_count()} host cores available"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
