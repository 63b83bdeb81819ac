#!/usr/bin/env python
"""bench.py -- rasterized Gaussians/s, forward+backward (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|cfg2|cfg5|long]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads (``--workload auto``: cfg2 at N = 1, cfg5 at N > 1):
  cfg2  BASELINE configs[1], the configuration the metric is quoted on: P = 200 000 static Gaussians, one 640x480 view. A step = one
        GaussianRasterizer forward + autograd backward (colour + depth cotangents), inputs resident in HBM. With N > 1 every rank
        renders its own keyframe of the same Gaussians and the gradients are summed with one RCCL all-reduce (weak scaling).
  cfg5  BASELINE configs[4], the mapping back-end's iteration that north_star shards: P = 2 000 000 Gaussians, 64 synthetic keyframes
        (SURVEY.md 8d poses) sharded round-robin over the N ranks -> 64 / N views per rank, gradients accumulated locally in a flat
        bucket, ONE all-reduce of 14 P floats (112 MB), then the fused Adam step on every replica. A step = that whole iteration;
        value = P * 64 * K / time (Gaussian-views per second, whole job); strong scaling (the 64 views are fixed).
  long  a SLAM-shaped variant of cfg2 (scale_mean 0.03 -> tile lists of several thousand entries, SH degree 3): profiles/ only.

One JSON line on rank 0 with `roofline` (dominant kernel = render_bwd, HIP events on its launch stream inside the timed region; `issue`
carries the binding roof: VALU issue slots used at the part's 2-cycle rate, and occupancy against what registers / LDS allow) and
`cpu_baseline` (the FASTEST CPU path on the box: the C restatement oracle/gs_oracle.c under OpenMP on all usable cores; the single-threaded
C port and BASELINE.md's baseline A -- the tile-binned pure-PyTorch CPU rasterizer oracle/torch_raster.py -- ride along inside it). The oracle
package is imported ONLY for that leg, outside the timed region. `m16`: the same workload at SH degree 3 (SURVEY.md 8d's second variant). At N = 1 the line also carries a short cfg5 measurement (`config5`), at N > 1 a short weak-scaling
cfg2 measurement (`weak_200k`), rank 0's single-GPU time for the same cfg5 iteration (`n1_reference`) with `speedup_vs_n1` / `efficiency`
derived from it, the ranks that took part (`ranks_seen`), the all-reduce alone (`allreduce_ms`) and the step with the exchange in two
overlapped pieces (`two_piece_exchange_ms_per_step`). The N > 1 lines name their metric "Gaussian-views/s": a different workload and
unit than the N = 1 headline, whose same-workload point is `config5` (N = 1) / `n1_reference` (N > 1).
"""
import argparse
import json
import os
import statistics
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

WIDTH, HEIGHT = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CFG2_P, CFG5_P, CFG5_KEYFRAMES = 200_000, 2_000_000, 64


def algorithmic_bytes(P, V, R, N, M):
    """minimum traffic of the reference algorithm, each item moved once, whole fwd+bwd and the
    render-backward kernel's share (per instance: 44 B read + 40 B accumulated; per pixel: 24 B read)."""
    total = 60 * P + (290 + 36 * M) * V + 292 * R + 52 * N
    render_bwd = 84 * R + 24 * N
    return total, render_bwd


class Scene:
    """Device-resident synthetic scene + per-keyframe rasterizer settings."""

    def __init__(self, P, dev, sh_degree=0, scale_mean=0.005, keyframes=(0,), cot_seed=1):
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from synthetic_scene import keyframe_pose, make_camera, make_cotangents, make_gaussians
        self.P, self.dev, self.sh_degree = P, dev, sh_degree
        self.g = make_gaussians(P, make_camera(WIDTH, HEIGHT), seed=0, sh_degree=sh_degree, scale_mean=scale_mean)
        T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)
        g = self.g
        self.means3D, self.shs, self.opac = T(g["means3D"], True), T(g["shs"], True), T(g["opacities"], True)
        self.scales, self.rots = T(g["scales"], True), T(g["rotations"], True)
        self.params = [self.means3D, self.shs, self.opac, self.scales, self.rots]      # the rasterizer's gradient-block order
        self.theta, self.rho = T(np.zeros(3), True), T(np.zeros(3), True)
        self.means2D = torch.zeros_like(self.means3D, requires_grad=True)             # the caller's screen-space gradient holder
        self.cams, self.rast = {}, {}
        for k in keyframes:
            R_w, t_w = keyframe_pose(k)
            cam = make_camera(WIDTH, HEIGHT, R=R_w, t=t_w)
            rs = GaussianRasterizationSettings(
                image_height=HEIGHT, image_width=WIDTH, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0,
                viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw), sh_degree=sh_degree,
                campos=T(cam.campos), prefiltered=False, debug=False)
            self.cams[k], self.rast[k] = cam, GaussianRasterizer(rs)
        gc, gd = make_cotangents(self.cams[keyframes[0]], seed=cot_seed)
        self.gc, self.gd, self.gcol, self.gdep = gc, gd, T(gc), T(gd)

    def fwd_bwd(self, k):
        color, radii, depth, opacity, n_touched = self.rast[k](means3D=self.means3D, means2D=self.means2D, opacities=self.opac, shs=self.shs,
                                                               scales=self.scales, rotations=self.rots, theta=self.theta, rho=self.rho)
        torch.autograd.backward([color, depth], [self.gcol, self.gdep])
        return radii

    def view_facts(self, k):
        """(visible, instances) of keyframe k, from the run itself."""
        from diff_gaussian_rasterization import _C
        rs = self.rast[k].raster_settings
        with torch.no_grad():
            nr, color, radii, *_ = _C.rasterize_gaussians(rs.bg, self.means3D, torch.Tensor([]), self.opac, self.scales, self.rots, 1.0, torch.Tensor([]),
                                                          rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, rs.tanfovx, rs.tanfovy, HEIGHT, WIDTH,
                                                          self.shs, self.sh_degree, rs.campos, False, False)
        return int((radii > 0).sum().item()), int(nr)


class RawScene:
    """The same synthetic scene held as the GaussianModel's RAW parameters (log-scales, unnormalised quaternions, logit opacities, SH DC),
    rendered through the multi-view entry point (gsr_forward_views / gsr_backward_views: one launch per pipeline stage for up to
    views.MAX_VIEWS keyframes) -- what BackEnd.map_static / map call per mapping iteration. Same images and gradient sums as Scene's
    view-by-view calls up to the activations' chain rules being applied inside the kernels."""

    def __init__(self, P, dev, scale_mean=0.005, keyframes=(0,), cot_seed=1, views_per_call=8):
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        from synthetic_scene import keyframe_pose, make_camera, make_cotangents, make_gaussians
        self.P, self.dev, self.sh_degree, self.views_per_call = P, dev, 0, views_per_call
        g = self.g = make_gaussians(P, make_camera(WIDTH, HEIGHT), seed=0, sh_degree=0, scale_mean=scale_mean)
        T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)
        op = np.clip(g["opacities"].astype(np.float64), 1e-6, 1 - 1e-6)
        self.xyz, self.f_dc = T(g["means3D"], True), T(g["shs"][:, :1, :], True)
        self.opacity, self.scaling, self.rotation = T(np.log(op / (1 - op)), True), T(np.log(g["scales"]), True), T(g["rotations"], True)
        self.params = [self.xyz, self.f_dc, self.opacity, self.scaling, self.rotation]          # the optimizer's order (GM:404-434)
        self.bg = T([1.0, 1.0, 1.0])
        self.cams, self.settings, self.poses = {}, {}, {}
        for k in keyframes:
            R_w, t_w = keyframe_pose(k)
            cam = make_camera(WIDTH, HEIGHT, R=R_w, t=t_w)
            self.cams[k] = cam
            self.settings[k] = GaussianRasterizationSettings(
                image_height=HEIGHT, image_width=WIDTH, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=self.bg, scale_modifier=1.0,
                viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw), sh_degree=0,
                campos=T(cam.campos), prefiltered=False, debug=False)
            self.poses[k] = (T(np.zeros(3), True), T(np.zeros(3), True))
        gc, gd = make_cotangents(self.cams[keyframes[0]], seed=cot_seed)
        self.gcol, self.gdep = T(gc), T(gd)
        self.batched_calls = 0

    def fwd_bwd_views(self, ks):
        from diff_gaussian_rasterization import _C, views
        for lo in range(0, len(ks), self.views_per_call):
            part = ks[lo:lo + self.views_per_call]
            block = torch.zeros((len(part), self.P, 3), device=self.dev)
            pts = [block[v].requires_grad_(True) for v in range(len(part))]
            outs = views.rasterize_views_raw([self.settings[k] for k in part], self.xyz, pts, self.scaling, self.rotation, self.opacity, self.f_dc,
                                             None, poses=[self.poses[k] for k in part])
            tensors, grads = [], []
            for o in outs:
                tensors += [o[0], o[2]]
                grads += [self.gcol, self.gdep]
            torch.autograd.backward(tensors, grads)
            for k in part:
                self.poses[k][0].grad = self.poses[k][1].grad = None
        self.batched_calls = _C.set_option("views_batched")

    def view_facts(self, k):
        from diff_gaussian_rasterization import raw
        with torch.no_grad():
            out = raw.rasterize_gaussians_raw(self.settings[k], self.xyz.detach(), torch.zeros_like(self.xyz), self.scaling.detach(), self.rotation.detach(),
                                              self.opacity.detach(), self.f_dc.detach(), None, None, None, None, None, None, None)
            torch.cuda.synchronize(self.dev)
        from diff_gaussian_rasterization import _C
        return int((out[1] > 0).sum().item()), int(_C.forward_status()[1])


STEP_KERNEL_SOURCES = ("gs_device.h", "gs_forward.h", "gs_render.h", "gs_backward.h", "gs_views.h")


def csrc_digest():
    """sha256 over the device sources of the step's kernels (4dgs-slam_amd/csrc/: preprocess / binning / sort / render / backward bodies and
    their multi-view wrappers): identifies the kernels a committed PMC traffic figure belongs to. The other headers (losses, control nodes,
    HexPlane, SLAM map kernels) and the host side in gs_capi.hip do not change what a launch of these kernels moves."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for name in STEP_KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kernel="render_bwd"):
    """HBM traffic per launch of the dominant kernel from the newest committed PMC collection (profiles/r*_hbm_traffic.json; a run under
    rocprofv3 --pmc cannot be part of a timed bench) -- used ONLY if the file was collected on exactly these kernel sources."""
    import glob
    digest = csrc_digest()
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        src = {"file": os.path.relpath(f, REPO), "collected_at_head": d.get("head"), "collected_on_csrc_sha256": d.get("csrc_sha256"), "bench_csrc_sha256": digest}
        if d.get("csrc_sha256") == digest:
            return d.get(f"{kernel}_bytes_per_launch"), src
        src["refused"] = "collected on different kernel sources: not quoted"
        return None, src
    return None, {"file": None, "bench_csrc_sha256": digest}


def committed_issue():
    """What actually bounds the two tile kernels -- instruction issue, not HBM -- from the newest committed counter collection
    (profiles/r*_tile_kernel_counters.json: SQ passes + rocprofv3's derived metrics of this same workload), quoted under the same digest rule
    as the traffic figure. Headline pair per kernel: `valu_frac` -- VALU wave-instructions x 2 cycles against the kernel's
    cycles, i.e. priced at the part's rate -- and `occupancy_frac` -- mean resident waves per SIMD over what registers / LDS allow."""
    import glob
    digest = csrc_digest()
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_tile_kernel_counters.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        src = {"file": os.path.relpath(f, REPO), "collected_on_csrc_sha256": d.get("csrc_sha256"), "bench_csrc_sha256": digest}
        if d.get("csrc_sha256") != digest:
            src["refused"] = "collected on different kernel sources: not quoted"
            return {"source": src}
        out = {"source": src, "bound": "SIMD instruction issue: a wave64 VALU instruction occupies its SIMD for 2 cycles (MI355X_MICROARCH.md), so the roof is "
                                       "kernel_cycles / 2 VALU wave-instructions per SIMD -- reached only with >= 4 ready waves per SIMD, because a single wave "
                                       "issues one instruction per ~8 cycles (profiles/r02_ubench_issue.json)",
               "how_to_read": "valu_frac = 2 cycles x SQ_INSTS_VALU / 1024 SIMDs / kernel cycles: the share of the VALU pipe's issue slots the launch uses, priced at "
                              "the PART's rate whatever the occupancy was (SALU and LDS instructions issue on their own ports and are listed, not added). "
                              "occupancy_frac = mean resident waves per SIMD / the waves the kernel's registers and LDS allow: what tails, prologues and "
                              "uneven tiles leave unused. issue_frac_at_occupancy (rounds 4-5's `issue_frac`) prices an instruction at the cycles a SIMD needs "
                              "at the launch's OWN mean occupancy: low occupancy lowers that roof instead of the fraction, so it is not an independent roof"}
        LDS_PER_CU, REGS_PER_LANE = 160 * 1024, 512
        for k in ("render_fwd", "render_bwd"):
            e = d["kernels"].get(k, {})
            der = e.get("derived", {})
            ipw = der.get("instr_per_wave", {})
            vg, lds, wg = e.get("_VGPR_Count"), e.get("_LDS_Block_Size"), e.get("_Workgroup_Size") or 256
            allowed = None
            if vg:
                # (rocprofv3's kernel trace reports HALF the per-lane register count on gfx950: 40 / 44 for the 80 / 88 registers that
                # hipcc -Rpass-analysis=kernel-resource-usage prints for these two kernels)
                by_regs = min(8, REGS_PER_LANE // (-(-2 * int(vg) // 8) * 8))
                by_lds = (LDS_PER_CU // int(lds)) * (int(wg) // 64) // 4 if lds else 8
                allowed = max(1, min(by_regs, by_lds, 8))
            occ = der.get("mean_waves_per_simd")
            out[k] = {"valu_frac": der.get("valu_busy_2cycle_view"), "mean_waves_per_simd": occ, "waves_per_simd_allowed": allowed,
                      "occupancy_frac": (occ / allowed) if occ and allowed else None,
                      "VALUBusy_percent_gfx94x_formula": e.get("VALUBusy"), "OccupancyPercent": e.get("OccupancyPercent"),
                      "wave_instructions_per_wave": {n: ipw.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")},
                      "waves": e.get("SQ_WAVES")}
        # rounds 4-5's figure, kept under its honest name: wave-instructions per SIMD x the cycles one of them costs a SIMD at the MEASURED occupancy
        # (profiles/r02_ubench_issue.json, independent v_fma_f32: 8.5 / 4.25 / 3.0 / 2.42 cycles at 1 / 2 / 3 / 4 waves per SIMD, interpolated) / the
        # kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs)
        try:
            ub = json.load(open(os.path.join(REPO, "profiles", "r02_ubench_issue.json")))
            cpi = next(r for r in ub["results"] if r["op"].startswith("v_fma_f32 independent"))["cycles_per_wave_instruction_per_simd"]
            table = sorted((float(k.split()[0]), float(v)) for k, v in cpi.items())
            def cycles_per_instruction(occ):
                occ = min(max(occ, table[0][0]), table[-1][0])
                for (x0, y0), (x1, y1) in zip(table, table[1:]):
                    if x0 <= occ <= x1:
                        return y0 + (y1 - y0) * (occ - x0) / (x1 - x0)
                return table[-1][1]
            for k in ("render_fwd", "render_bwd"):
                e = d["kernels"].get(k, {})
                n_instr = sum(e.get(c, 0) or 0 for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
                cycles = (e.get("GRBM_GUI_ACTIVE") or 0) / 8.0
                occ = out[k].get("mean_waves_per_simd")
                if n_instr and cycles and occ:
                    c = cycles_per_instruction(float(occ))
                    out[k]["issue_frac_at_occupancy"] = n_instr / 1024.0 * c / cycles
                    out[k]["issue_frac_at_occupancy_inputs"] = {"wave_instructions_per_simd": n_instr / 1024.0, "cycles_per_wave_instruction_at_occupancy": c, "kernel_cycles": cycles}
        except Exception:
            pass
        if out.get("render_bwd", {}).get("valu_frac") is not None:
            out["frac"] = out["render_bwd"]["valu_frac"]            # roofline.issue.frac: the dominant kernel's share of the VALU issue roof
            out["occupancy_frac"] = out["render_bwd"]["occupancy_frac"]
        # (quadrant, Gaussian) pairs per wave from the cycle-accounting build of the same sources, when it was collected with them
        pc = f.replace("_tile_kernel_counters.json", "_phase_cycles.json")
        try:
            ph = json.load(open(pc))
            if ph.get("csrc_sha256") == digest:
                for k, key in (("render_fwd", "pairs"), ("render_bwd", "candidate_pairs")):
                    pairs = ph[k][key]
                    ipw = out[k]["wave_instructions_per_wave"]
                    out[k]["pairs_per_wave"] = pairs
                    out[k]["wave_instructions_per_pair"] = sum(v for v in ipw.values() if v) / pairs if pairs else None
        except Exception:
            pass
        return out
    return {"source": {"file": None, "bench_csrc_sha256": digest}}


def note(msg):
    """progress on stderr (the JSON line on stdout stays the only stdout output)"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def timed(step, steps, warmup, barrier):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def run_cfg2(scene, world, rank, steps, warmup, barrier, k):
    """One view per rank (+ all-reduce when world > 1). Returns (seconds for `steps`, all-reduce mode)."""
    from mapping_shard import GradBucket
    bucket = GradBucket(scene.params) if world > 1 else None
    mode = {}

    def step():
        for p_ in scene.params + [scene.theta, scene.rho, scene.means2D]:
            p_.grad = None
        scene.fwd_bwd(k)
        if bucket is not None:
            mode["allreduce"] = bucket.all_reduce_grads()   # in place on the backward's own output range when possible
    return timed(step, steps, warmup, barrier), mode.get("allreduce"), step


def graph_replayed_step(scene, k, steps):
    """The SAME forward + backward captured once as a hipGraph -- the rasterizer's lazy mode, no host wait inside the step (what the SLAM
    front-end's tracking loop runs, slam/tracking_graph.py) -- and replayed `steps` times: the step with the host out of the loop.
    Reported beside the headline (whose eager step depends on the box's host speed), never instead of it. Returns a dict or None."""
    from diff_gaussian_rasterization import _C
    leaves = scene.params + [scene.theta, scene.rho, scene.means2D]
    lazy_before = _C.set_option("lazy", 1)
    try:
        side = torch.cuda.Stream(device=scene.dev)
        side.wait_stream(torch.cuda.current_stream(scene.dev))
        with torch.cuda.stream(side):
            for _ in range(3):                      # allocator warm-up + the capacity estimate lazy mode sizes its buffers from
                for p_ in leaves:
                    p_.grad = None
                scene.fwd_bwd(k)
        torch.cuda.current_stream(scene.dev).wait_stream(side)
        for p_ in leaves:
            p_.grad = None
        overflow_before = _C.forward_status()[0]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            scene.fwd_bwd(k)
    except Exception as exc:                        # a capture that fails must not cost the bench its line
        note(f"graph-replayed step skipped: {type(exc).__name__}: {exc}")
        return None
    finally:
        _C.set_option("lazy", lazy_before)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize(scene.dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize(scene.dev)
    dt = (time.perf_counter() - t0) / steps
    ok = _C.forward_status()[0] == overflow_before  # no replay outgrew its (captured) binning capacity
    for p_ in leaves:
        p_.grad = None
    del graph
    return {"ms_per_step": dt * 1e3, "value": scene.P / dt, "unit": "Gaussians/s", "steps": steps, "overflow_free": bool(ok),
            "what": "the same forward + backward captured as ONE hipGraph (lazy forward: no host wait) and replayed: host out of the loop"}


def make_cfg5(scene, keyframes, overlap=False, local=False, exchange="all_reduce"):
    """The config #5 iteration. local=True: every keyframe on THIS rank, no collective (the single-GPU point of the scaling curve).
    A RawScene goes through the multi-view entry point (this rank's keyframes in groups of 8 per call), a Scene view by view."""
    from fused_adam import FusedAdam
    from mapping_shard import ShardedMappingStep
    opt = FusedAdam([{"params": [p], "lr": 0.0, "name": n} for p, n in zip(scene.params, ("xyz", "f", "opacity", "scaling", "rotation"))],
                    lr=0.0, eps=1e-15)      # lr = 0: the full Adam arithmetic runs, the scene (hence the workload) stays put
    multi = isinstance(scene, RawScene)
    sms = ShardedMappingStep(scene.params, keyframes, None if multi else (lambda k: scene.fwd_bwd(k)), optimizer=opt, overlap=overlap,
                             exchange=exchange, views_fn=(lambda ks: scene.fwd_bwd_views(ks)) if multi else None, local=local)

    def step():
        if not multi:
            scene.theta.grad = scene.rho.grad = scene.means2D.grad = None
        sms.step()
    return sms, step


def usable_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the machine's
    count even inside a container limited to a few CPUs; 256 spinning OpenMP threads on 8 usable cores do not finish)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baselines_in_child(P, sh_degree, scale_mean, timeout_s):
    """Runs the CPU-baseline leg in a child process with a hard wall-clock limit, so that a slow host can never stall the bench."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--gaussians", str(P), "--sh-degree", str(sh_degree),
           "--scale-mean", str(scale_mean)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, OMP_WAIT_POLICY="PASSIVE"))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            return d["cpu_baseline"], d["cpu_baseline_port"]
        why = f"child exited with {r.returncode}: {r.stderr[-300:]}"
    except subprocess.TimeoutExpired:
        why = f"did not finish within {timeout_s} s on this host ({usable_cores()} usable cores of {os.cpu_count()})"
    return {"value": None, "unit": "Gaussians/s", "cores": usable_cores(), "kind": "port", "sample": "not measured: " + why}, None


def cpu_baselines(scene_g, cam, gc, gd, P, sh_degree, budget_s):
    """BASELINE.md baseline A (tile-binned pure-PyTorch CPU rasterizer) on all cores and on one, plus the C port (one core; OpenMP)."""
    import oracle  # CPU baseline leg only (test infrastructure; never in the product path)
    from oracle import torch_raster
    cores = usable_cores()
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), requires_grad=rg)
    gx, gy = (WIDTH + 15) // 16, (HEIGHT + 15) // 16
    win = (gx // 2 - 5, gy // 2 - 3, gx // 2 + 5, gy // 2 + 3)            # 10 x 6 = 60 of the 1200 tiles, centred
    n_win = (win[2] - win[0]) * (win[3] - win[1])

    def torch_once(window):
        leaves = dict(means3D=T(scene_g["means3D"], True), opacities=T(scene_g["opacities"], True), shs=T(scene_g["shs"], True),
                      scales=T(scene_g["scales"], True), rotations=T(scene_g["rotations"], True))
        m2d = torch.zeros(P, 3, requires_grad=True)
        t0 = time.perf_counter()
        out = torch_raster.rasterize(leaves["means3D"], m2d, leaves["opacities"], shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"],
                                     bg=torch.ones(3), viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), campos=T(cam.campos),
                                     tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=HEIGHT, image_width=WIDTH, sh_degree=sh_degree,
                                     tile_window=window)
        if out[0].requires_grad:                                         # (no tile composited: nothing to back-propagate)
            torch.autograd.backward([out[0], out[2]], [T(gc), T(gd)])
        return time.perf_counter() - t0

    def sample(threads, reps):
        torch.set_num_threads(threads)
        t_pre = min(torch_once((0, 0, 0, 0)) for _ in range(2))              # per-Gaussian part + tile bookkeeping of ALL tiles, no tile composited
        t_win = statistics.median(torch_once(win) for _ in range(reps))
        full = t_pre + (t_win - t_pre) * (gx * gy) / n_win                   # the composited tiles scale linearly with their number
        return {"value": P / full, "seconds_full_extrapolated": full, "seconds_window": t_win, "seconds_no_tiles": t_pre}

    prev = torch.get_num_threads()
    note(f"cpu baseline A: torch tile-binned rasterizer, {cores} threads")
    allc = sample(cores, 3)
    note(f"  all cores: {allc['seconds_window']:.2f} s per window sample -> {allc['seconds_full_extrapolated']:.1f} s estimated for the frame")
    # the whole frame, measured (not extrapolated), when the estimate says it fits the budget; the window estimate stays as a cross-check
    full_measured = None
    if allc["seconds_full_extrapolated"] < 0.6 * budget_s:
        torch.set_num_threads(cores)
        full_measured = torch_once(None)
        note(f"  full frame measured: {full_measured:.2f} s")
    one = sample(1, 3) if allc["seconds_window"] * 6 + (full_measured or 0) < budget_s else None
    torch.set_num_threads(prev)
    note("cpu baseline port: C oracle, 1 thread, then OpenMP")

    def port(variant, threads):
        oracle.set_variant(variant, threads)
        t1 = time.perf_counter()
        o, st = oracle.rasterize_forward(bg=np.ones(3, np.float32), means3D=scene_g["means3D"], opacities=scene_g["opacities"], shs=scene_g["shs"],
                                         scales=scene_g["scales"], rotations=scene_g["rotations"], viewmatrix=cam.viewmatrix,
                                         projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                         image_height=HEIGHT, image_width=WIDTH, sh_degree=sh_degree)
        oracle.rasterize_backward(st, projmatrix_raw=cam.projmatrix_raw, dL_dcolor=gc, dL_ddepth=gd)
        return time.perf_counter() - t1

    t_port1 = port("serial", 1)
    t_omp = min(port("omp", cores) for _ in range(2))
    oracle.set_variant("serial")
    what = (f"oracle/torch_raster.py (tile-binned pure-PyTorch CPU rasterizer = BASELINE.md baseline A; this repo's restatement, the reference has "
            f"no CPU path), fwd+bwd of the full {P} Gaussians @{WIDTH}x{HEIGHT}, torch.set_num_threads({cores}): ")
    if full_measured is not None:
        value = P / full_measured
        what += (f"ONE full frame measured ({full_measured:.2f} s, all {gx * gy} tiles composited and back-propagated); cross-check: a centred window of "
                 f"{n_win} tiles extrapolated linearly in the tile count gives {allc['seconds_full_extrapolated']:.1f} s")
    else:
        value = allc["value"]
        what += (f"per-Gaussian stage over all P, then a centred window of {n_win} of {gx * gy} tiles composited and back-propagated, median of 3, "
                 f"EXTRAPOLATED linearly in the tile count ({allc['seconds_window']:.2f} s per sample -> {allc['seconds_full_extrapolated']:.1f} s for "
                 f"the frame: the full frame did not fit the time budget on this host)")
    torch_line = {"value": value, "unit": "Gaussians/s", "cores": cores, "kind": "port", "sample": what,
                  "seconds_full_frame_measured": full_measured, "seconds_full_frame_window_estimate": allc["seconds_full_extrapolated"],
                  "torch_one_core": None if one is None else {"value": one["value"], "cores": 1, "seconds_full_extrapolated": one["seconds_full_extrapolated"]}}
    # `cpu_baseline` is the FASTEST CPU path on this box: the C restatement under OpenMP on all usable cores; the PyTorch tile-binned
    # rasterizer (BASELINE.md's baseline A, what north_star's ">= 5x the host-CPU PyTorch fallback" refers to) and the single-threaded C port ride along
    base = {"value": P / t_omp, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"oracle/gs_oracle.c built with OpenMP (libgs_oracle_omp.so: tile and per-Gaussian loops parallel; binning, key sort and preprocess "
                      f"serial), ONE full fwd+bwd of the same workload ({P} Gaussians @{WIDTH}x{HEIGHT}), best of 2: {t_omp:.3f} s on {cores} threads -- the fastest "
                      f"of the three CPU paths measured here",
            "seconds": t_omp,
            "c_port_one_thread": {"value": P / t_port1, "cores": 1, "seconds": t_port1, "sample": "oracle/gs_oracle.c, single-threaded, one full fwd+bwd"},
            "torch_fallback": torch_line}
    port_line = {"value": P / t_port1, "unit": "Gaussians/s", "cores": 1, "kind": "port",
                 "sample": f"oracle/gs_oracle.c, one full fwd+bwd of the same workload ({t_port1:.2f} s), single-threaded",
                 "openmp_all_cores": {"value": P / t_omp, "cores": cores, "seconds": t_omp,
                                      "note": "libgs_oracle_omp.so: tile and per-Gaussian loops under OpenMP; binning, key sort and preprocess stay serial"}}
    return base, port_line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", choices=["auto", "cfg2", "cfg5", "long"])
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--keyframes", type=int, default=CFG5_KEYFRAMES)
    ap.add_argument("--sh-degree", type=int, default=None)
    ap.add_argument("--scale-mean", type=float, default=None)
    ap.add_argument("--exchange", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="cfg5: one all-reduce of the gradient bucket (default) or reduce-scatter -> Adam on the local slice -> all-gather of parameters")
    ap.add_argument("--view-by-view", action="store_true", help="cfg5: one rasterizer call per keyframe (round 3's step) instead of the multi-view entry point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the embedded secondary measurements (config5 / weak_200k / n1_reference)")
    ap.add_argument("--graph-replay", action="store_true", help="with --no-secondary: still measure the step captured as one hipGraph (tools/dev_ab.sh)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-timeout", type=float, default=150.0)
    args = ap.parse_args()
    if args.cpu_baseline_child:        # CPU only: no GPU, no product code
        from synthetic_scene import make_camera, make_cotangents, make_gaussians
        cam = make_camera(WIDTH, HEIGHT)
        P = args.gaussians or CFG2_P
        g = make_gaussians(P, cam, seed=0, sh_degree=args.sh_degree or 0, scale_mean=args.scale_mean or 0.005)
        gc, gd = make_cotangents(cam, seed=1)
        a, b = cpu_baselines(g, cam, gc, gd, P, args.sh_degree or 0, budget_s=90.0)
        print(json.dumps({"cpu_baseline": a, "cpu_baseline_port": b}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # `python bench.py --gpus N`: become the launcher -- N ranks of this same command under torch.distributed.run, one per GPU; rank 0's JSON
        # line is the only stdout output of the children, and it is passed through unchanged
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        note(f"self-launch: {' '.join(cmd)}")
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
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

    from diff_gaussian_rasterization import _C

    workload = args.workload if args.workload != "auto" else ("cfg2" if world == 1 else "cfg5")
    sh_degree = args.sh_degree if args.sh_degree is not None else (3 if workload == "long" else 0)
    scale_mean = args.scale_mean if args.scale_mean is not None else (0.03 if workload == "long" else 0.005)
    N = WIDTH * HEIGHT

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    def kernel_breakdown(step, reps=5):
        """per-kernel time (HIP events on the launch stream) in a separate, untimed pass"""
        _C.profile_reset()
        _C.profile_enable(True)
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        _C.profile_enable(False)
        return {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in _C.profile_read().items() if v[1]}  # us per launch

    out = None
    if workload in ("cfg2", "long"):
        P = args.gaussians or CFG2_P
        scene = Scene(P, dev, sh_degree, scale_mean, keyframes=(rank,), cot_seed=1 + rank)
        note(f"{workload}: P = {P}, world = {world}")
        # headline: speculative binning (the SLAM loop's steady state); the dominant kernel is timed inside the timed region
        _, _, step = run_cfg2(scene, world, rank, 0, args.warmup, barrier, rank)
        _C.profile_reset()
        _C.profile_enable(["render_bwd"])       # 2 event records per step on the launch stream
        dt = reduce_max(timed(step, args.steps, 0, barrier))
        _C.profile_enable(False)
        dom_ms, dom_calls = _C.profile_read()["render_bwd"]
        kern = kernel_breakdown(step)
        note(f"timed region done: {dt / args.steps * 1e3:.4f} ms per step")
        nospec = None
        if world == 1:
            # the same step with the speculation off: the host waits for num_rendered before it allocates and enqueues the binning,
            # as the reference does (rasterizer_impl.cu:283-284) -- the cost of a frame whose size cannot be predicted
            _C.set_option("speculate", 0)
            nospec = timed(step, max(args.steps // 4, 5), 5, barrier) / max(args.steps // 4, 5)
            _C.set_option("speculate", 1)
        if rank == 0:
            V, nr = scene.view_facts(rank)
            M = int(scene.shs.shape[1])
            b_total, b_dom = algorithmic_bytes(P, V, nr, N, M)
            dom_s = dom_ms / max(dom_calls, 1) * 1e-3
            achieved = b_dom / dom_s / 1e9 if dom_s > 0 else 0.0
            traffic, traffic_source = (committed_traffic("render_bwd") if workload == "cfg2" and P == CFG2_P else (None, None))
            out = {
                "metric": "rasterized Gaussians/s fwd+bwd @640x480 (200k G)", "value": world * P * args.steps / dt, "unit": "Gaussians/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": (f"configs[1]: {P} static Gaussians, 1 cam @{WIDTH}x{HEIGHT} per GPU, SH degree {sh_degree}, fwd+bwd" if workload == "cfg2" else
                                        f"SLAM-shaped variant: {P} Gaussians, scale_mean {scale_mean}, SH degree {sh_degree}, 1 cam @{WIDTH}x{HEIGHT}, fwd+bwd")
                                       + (f", {world} views sharded + RCCL all-reduce of gradients" if world > 1 else ""),
                           "visible": V, "instances": nr, "pixels": N, "host_binding": _C.binding()},
                "roofline": {"bound": "hbm", "kernel": "render_bwd", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                             "algorithmic_bytes_per_launch": b_dom, "kernel_us": dom_s * 1e6,
                             "whole_step_algorithmic_bytes": b_total, "whole_step_GBps": b_total / (dt / args.steps) / 1e9,
                             # nominal: every instance x the 256 pixels of its tile; the kernel EVALUATES the (quadrant, Gaussian) pairs its ballots
                             # let through (issue.render_bwd.pairs_per_wave x waves x 64 lanes: ~37 % of the nominal count at config #2)
                             "nominal_pair_evals_per_s_bwd": nr * 256 / dom_s if dom_s > 0 else None,
                             "issue": committed_issue() if workload == "cfg2" and P == CFG2_P else None,
                             "note": "the tile kernels are instruction-issue bound, not HBM bound (DESIGN.md 4; `issue` carries the counters that say so): frac is "
                                     "reported against the HBM roof because north_star asks for it"},
                "kernel_us": kern,
            }
            iss = out["roofline"].get("issue") or {}
            rb = iss.get("render_bwd") or {}
            if rb.get("pairs_per_wave") and rb.get("waves") and dom_s > 0:
                out["roofline"]["evaluated_pair_evals_per_s_bwd"] = rb["pairs_per_wave"] * rb["waves"] * 64 / dom_s
            if nospec is not None:
                out["ms_per_step_nonspeculative"] = nospec * 1e3
            if world == 1 and (not args.no_secondary or args.graph_replay):
                replayed = graph_replayed_step(scene, rank, max(args.steps, 20))
                if replayed is not None:
                    out["graph_replay"] = replayed
        # ---- secondary: SURVEY.md 8(d)'s M = 16 variant of the headline (SH degree 3): the one place where kernels of the step are bandwidth-bound ----
        if world == 1 and workload == "cfg2" and P == CFG2_P and sh_degree == 0 and not args.no_secondary and rank == 0:
            note("secondary: the headline workload at SH degree 3 (M = 16)")
            s16 = Scene(P, dev, 3, scale_mean, keyframes=(rank,), cot_seed=1 + rank)
            _, _, step16 = run_cfg2(s16, 1, 0, 0, 10, barrier, rank)
            n16 = max(args.steps // 4, 20)
            t16 = timed(step16, n16, 0, barrier) / n16
            k16 = kernel_breakdown(step16)
            V16, nr16 = s16.view_facts(rank)
            b16, _ = algorithmic_bytes(P, V16, nr16, N, 16)
            # what the SH coefficients add: 48 floats read by preprocess_fwd (visible Gaussians), 48 read + 48 written by geometry_bwd
            sh_bytes = (48 * 4) * V16, (48 * 4) * (V16 + P)
            out["m16"] = {"workload": f"configs[1] with SH degree 3 (M = 16): {P} Gaussians, 1 cam @{WIDTH}x{HEIGHT}, fwd+bwd", "ms_per_step": t16 * 1e3,
                          "value": P / t16, "unit": "Gaussians/s", "steps": n16, "kernel_us": k16, "whole_step_algorithmic_bytes": b16,
                          "coefficient_bytes": {"preprocess_fwd": sh_bytes[0], "geometry_bwd": sh_bytes[1]},
                          "coefficient_GBps": {"preprocess_fwd": sh_bytes[0] / (k16.get("preprocess_fwd", 0) * 1e-6) / 1e9 if k16.get("preprocess_fwd") else None,
                                               "geometry_bwd": sh_bytes[1] / (k16.get("geometry_bwd", 0) * 1e-6) / 1e9 if k16.get("geometry_bwd") else None},
                          "note": "coefficient_GBps divides the SH coefficients' bytes alone by the whole kernel's time: a lower bound of what the kernel moves"}
            del s16
            torch.cuda.empty_cache()
        # ---- secondary: a short config #5 iteration on this one GPU (so that the N > 1 lines have a same-workload N = 1 point) ----
        if world == 1 and workload == "cfg2" and not args.no_secondary:
            note("secondary: config #5 iteration on one GPU")
            del scene
            torch.cuda.empty_cache()
            kfs5 = list(range(args.keyframes))
            s5 = RawScene(CFG5_P, dev, 0.005, keyframes=tuple(kfs5))
            sms, step5 = make_cfg5(s5, kfs5)
            t5 = timed(step5, 3, 2, barrier) / 3
            out["config5"] = {"workload": f"configs[4] on ONE GPU: {CFG5_P} Gaussians x {args.keyframes} keyframes through the multi-view entry point "
                                          f"({s5.views_per_call} views per call), gradients accumulated, fused Adam",
                              "ms_per_step": t5 * 1e3, "value": CFG5_P * args.keyframes / t5, "unit": "Gaussian-views/s", "steps": 3,
                              "multi_view_calls_batched": s5.batched_calls}
            del s5, sms
            torch.cuda.empty_cache()
            s5 = Scene(CFG5_P, dev, 0, 0.005, keyframes=tuple(kfs5))              # round 3's step beside it: one rasterizer call per keyframe
            sms, step5 = make_cfg5(s5, kfs5)
            t5v = timed(step5, 2, 1, barrier) / 2
            out["config5"]["view_by_view_ms_per_step"] = t5v * 1e3
            del s5, sms
            torch.cuda.empty_cache()
        # ---- secondary: BASELINE config #3 (500k Gaussians + the HexPlane deformation network, 8 keyframes, pose gradients) on this one GPU ----
        if world == 1 and workload == "cfg2" and not args.no_secondary:
            note("secondary: config #3 iteration (deformation network, 8 keyframes)")
            try:
                from tools.bench_config3 import measure as measure_config3
                c3 = measure_config3(["fused", "batched"], iters=3)
                out["config3"] = {"workload": "configs[2]: " + c3["workload"] + "; `batched`: the keyframes of the iteration through the deformation "
                                              "producer and the multi-view rasterizer at once (render_views(dynamic=True)), `per_view`: one render(dynamic=True) "
                                              "per keyframe (rounds 1-4)",
                                  "ms_per_step": c3["batched_ms_per_iteration"], "value": c3["batched_gaussian_views_per_s"], "unit": "Gaussian-views/s", "steps": 3,
                                  "per_view_ms_per_step": c3["fused_ms_per_iteration"], "peak_memory_GB": c3["peak_memory_GB"]}
            except Exception as e:      # the headline line must not depend on the secondary
                out["config3"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
    else:   # ---- cfg5 -------------------------------------------------------------------------------------------------------
        P = args.gaussians or CFG5_P
        kfs = list(range(args.keyframes))
        mine = tuple(kfs[rank::world]) if world > 1 else tuple(kfs)
        multi_view = not args.view_by_view and sh_degree == 0
        scene = RawScene(P, dev, scale_mean, keyframes=mine) if multi_view else Scene(P, dev, sh_degree, scale_mean, keyframes=mine)
        sms, step = make_cfg5(scene, kfs, exchange=args.exchange)
        for _ in range(args.warmup):
            step()
        barrier()
        _C.profile_reset()
        _C.profile_enable(["render_bwd"])
        dt = reduce_max(timed(step, args.steps, 0, barrier))
        _C.profile_enable(False)
        dom_ms, dom_calls = _C.profile_read()["render_bwd"]
        # what every rank holds after the timed steps (sum of |parameter|, float64): equal on all ranks if the exchange kept them in step, and --
        # same seeds, same step count -- equal between the all-reduce and the reduce-scatter exchange up to summation order
        param_digests = None
        if world > 1:
            mine_d = torch.stack([p.detach().double().abs().sum() for p in scene.params]).sum().reshape(1)
            every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(every, mine_d)
            param_digests = [float(t.item()) for t in every]
        kern = kernel_breakdown(step, reps=1)
        # all-reduce alone (gradients already in the bucket): exposed once per step
        ar_ms = None
        if world > 1:
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                sms.bucket.all_reduce_grads()
            barrier()
            ar_ms = (time.perf_counter() - t0) / 5 * 1e3
        # who took part (an all-gather of rank ids over RCCL), and the same step with the exchange in two pieces (mapping_shard: overlap)
        ranks_seen, two_piece_ms = [0], None
        if world > 1:
            ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
            ranks_seen = sorted(int(t.item()) for t in ids)
            if len(sms.keyframes) >= 2 and args.exchange == "all_reduce":
                sms2, step2 = make_cfg5(scene, kfs, overlap=True)
                two_piece_ms = reduce_max(timed(step2, max(2, args.steps // 2), 1, barrier)) / max(2, args.steps // 2) * 1e3
                del sms2
                sms.bucket.attach()
        # the other exchange beside the chosen one: reduce-scatter -> Adam on the local slice -> all-gather of the parameters
        other_ms, other = None, ("reduce_scatter" if args.exchange == "all_reduce" else "all_reduce")
        if world > 1 and not args.no_secondary:
            sms3, step3 = make_cfg5(scene, kfs, exchange=other)
            other_ms = reduce_max(timed(step3, max(2, args.steps // 2), 1, barrier)) / max(2, args.steps // 2) * 1e3
            del sms3
            if args.exchange == "all_reduce":
                sms.bucket.attach()
        facts = [scene.view_facts(k) for k in sms.keyframes]
        if rank == 0:
            Vm, Rm = sum(f[0] for f in facts) / len(facts), sum(f[1] for f in facts) / len(facts)
            M = int(scene.shs.shape[1]) if hasattr(scene, "shs") else 1
            b_total, b_dom = algorithmic_bytes(P, Vm, Rm, N, M)
            dom_s = dom_ms / max(dom_calls, 1) * 1e-3
            achieved = b_dom / dom_s / 1e9 if dom_s > 0 else 0.0
            out = {
                "metric": f"rasterized Gaussian-views/s fwd+bwd @640x480 ({P // 1000}k G x {len(kfs)} keyframes, view-sharded mapping iteration incl. all-reduce + Adam)",
                "value": P * len(kfs) * args.steps / dt, "unit": "Gaussian-views/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "ranks_seen": ranks_seen, "allreduce_ms": ar_ms, "two_piece_exchange_ms_per_step": two_piece_ms, "param_digests": param_digests,
                "exchange": args.exchange, f"{other}_exchange_ms_per_step": other_ms,
                "multi_view": bool(multi_view), "views_per_call": getattr(scene, "views_per_call", 1),
                "config": {"workload": f"configs[4]: {P} Gaussians, {len(kfs)} synthetic keyframes @{WIDTH}x{HEIGHT} sharded {world}-way ({len(sms.keyframes)} views per rank, "
                                       f"gradients accumulated locally), ONE all-reduce of {sms.bucket.nbytes} B, fused Adam step; value = Gaussian-views/s",
                           "views_per_rank": len(sms.keyframes), "allreduce_bytes": sms.bucket.nbytes, "allreduce_mode": sms.mode, "allreduce_ms": ar_ms,
                           "visible_mean": Vm, "instances_mean": Rm, "pixels": N, "host_binding": _C.binding()},
                "roofline": {"bound": "hbm", "kernel": "render_bwd", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_dom, "kernel_us": dom_s * 1e6,
                             "whole_step_algorithmic_bytes": b_total * len(sms.keyframes)},
                "kernel_us": kern,
            }
        if world > 1 and not args.no_secondary and args.gaussians is None:
            # the round-1 weak-scaling mode as a secondary figure: one 200k-Gaussian view per rank + all-reduce
            sw = Scene(CFG2_P, dev, 0, 0.005, keyframes=(rank,), cot_seed=1 + rank)
            dtw, mode_w, _ = run_cfg2(sw, world, rank, 20, 5, barrier, rank)
            dtw = reduce_max(dtw)
            if rank == 0:
                out["weak_200k"] = {"workload": f"{CFG2_P} Gaussians, one view per rank, all-reduce ({mode_w})", "ms_per_step": dtw / 20 * 1e3,
                                    "value": world * CFG2_P * 20 / dtw, "scaling": "weak"}
            del sw
        if world > 1 and not args.no_secondary:
            # rank 0 alone runs the whole 64-view iteration: the same-workload single-GPU point the scaling is measured against
            barrier()
            if rank == 0:
                del scene, sms
                torch.cuda.empty_cache()
                s1 = RawScene(P, dev, scale_mean, keyframes=tuple(kfs)) if multi_view else Scene(P, dev, sh_degree, scale_mean, keyframes=tuple(kfs))
                _, step1 = make_cfg5(s1, kfs, local=True)           # the same code path (multi-view calls, attached bucket, fused accumulation, fused Adam)
                step1()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    step1()
                torch.cuda.synchronize()
                t1 = (time.perf_counter() - t0) / 2
                out["n1_reference"] = {"ms_per_step": t1 * 1e3, "value": P * len(kfs) / t1, "note": "the same iteration on rank 0's GPU alone, no collective"}
                out["speedup_vs_n1"] = t1 * 1e3 / out["ms_per_step"]
                out["efficiency"] = out["speedup_vs_n1"] / world
            barrier()

    if rank == 0 and not args.no_cpu_baseline and world == 1 and workload in ("cfg2", "long"):
        out["cpu_baseline"], port = cpu_baselines_in_child(P, sh_degree, scale_mean, args.cpu_baseline_timeout)
        if port is not None:
            out["cpu_baseline_port"] = port
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
