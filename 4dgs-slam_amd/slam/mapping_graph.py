"""A run of STATIC mapping iterations as hipGraph replays (utils/slam_backend.py:1013-1224 is the loop being replaced):

    [schedule row -> keyframe slots] -> multi-view render (window keyframes + 2 random ones) -> fused mapping loss per view ->
    isotropic regulariser -> ONE backward -> densification statistics -> Adam on the six Gaussian groups -> gradient reset -> camera steps

Eager, that is ~55 launches driven by Python and autograd: 1.0-1.6 ms of wall time for 0.5 ms of device time at SLAM sizes. Captured, the host
issues one ``graph.replay()`` per iteration. What makes the iteration capturable (beyond what slam/tracking_graph.py already needed -- the
rasterizer's lazy mode, cameras in persistent device buffers, a camera step with its Adam counter on the device):

  * the two RANDOM keyframes of an iteration (:1031-1037) are rendered through two persistent *slots*: a slot is a camera object plus
    ground-truth / loss-weight images of its own, filled inside the graph by ``gsr_slot_gather`` from a device table of the candidate
    keyframes' buffer addresses and an index read from device memory -- which keyframes an iteration draws is an index write, not a capture;
  * everything that depends on the iteration NUMBER -- those indices, Adam's bias corrections, the position learning rate (GM:492-505) -- is a
    row of a device-side schedule written once per run; ``gsr_schedule_advance`` copies row ``counter`` into the "current" block the other
    kernels read and bumps the counter (include/slam_map.h), ``gsr_adam_step_scheduled`` reads its coefficients there (include/slam_losses.h);
  * the iterations that densify or reset opacities (every ``gaussian_update_every`` / ``gaussian_reset`` iterations) replace the model's
    tensors: they run eagerly (BackEnd._map_static_iteration) and end a run; the next run captures again.

The host draws the random keyframes for the whole run up front -- the same ``torch.randperm`` calls in the same order as the eager loop -- and
``iteration()`` is ONE code path: executed directly for the warm-up iterations of a run, captured once, then replayed. Results are
bit-identical to the eager loop (tests/test_hip_slam.py). A replayed forward pass that outgrows its (generously sized) binning buffer is
detected after the run through the sticky overflow counters; the run is then undone from a snapshot and repeated eagerly."""
import numpy as np
import torch

from diff_gaussian_rasterization import _C
import slam_losses

from . import _lib
from .camera import Camera

N_INDEX_WORDS = 2            # random keyframes per iteration (utils/slam_backend.py:1031-1037)
CAPTURE_MARGIN_PERMILLE = 500
CAPTURE_TILE_MARGIN_PERMILLE = 3000     # the longest tile list may grow 4x during the replays (see include/gs_rasterizer.h)
# initialize_map's runs: the instance count of the ONE view being fitted grew 1.82x inside a 99-iteration run right after the opacity reset
# (104 438 -> 190 427 instances at iteration 501 of the config #4 stand-in; 1.5x was the margin, 97 iterations were redone): 3x here
INIT_CAPTURE_MARGIN_PERMILLE = 2000


def device_store_budget(device, fraction, floor_bytes=256 << 20):
    """Bytes a per-keyframe store may hold on `device`: `fraction` of the memory that is free right now (never less than floor_bytes)."""
    try:
        free, _total = torch.cuda.mem_get_info(device)
    except Exception:
        return floor_bytes
    return max(int(free * fraction), floor_bytes)


class KeyframeOperands:
    """Per keyframe: the constant operands of its mapping loss (ground truth, weights), held so that their device addresses stay valid for
    the graphs that point at them (slam_losses keeps only a bounded cache).

    One entry per keyframe: the ground-truth image and depth ONCE, the loss weights per (rm_dynamic, dynamic) flag variant (a weight pair is
    2.4 MB at 640x480, the ground truth 4.9 MB -- three variants used to hold three copies of it once slam_losses' constants cache had
    evicted the keyframe). The store is bounded by BYTES -- BUDGET_FRACTION of the device memory free at its first use, least recently used
    keyframe first -- and follows Camera.clean() through slam_losses.drop_keyframe_constants. What is dropped is formed again on demand; a
    graph that still points at an entry's buffers holds the tensors itself."""

    BUDGET_FRACTION = 0.05

    def __init__(self):
        self._held = {}                 # id(viewpoint) -> [viewpoint, gt_image, gt_depth, {(rm_dynamic, dynamic): (w_rgb, w_depth, alpha)}, bytes]
        self._bytes, self._budget = 0, None
        slam_losses.on_drop_keyframe_constants(self.drop)

    @staticmethod
    def _nbytes(*tensors):
        return sum(t.numel() * t.element_size() for t in tensors if isinstance(t, torch.Tensor))

    def get(self, config, viewpoint, device, rm_dynamic=True, dynamic=False):
        """(gt_image, gt_depth, w_rgb, w_depth, alpha) of slam_losses.mapping_loss_operands, computed once per (keyframe, flags) and held:
        a keyframe's ground truth and masks never change. The eager loop's get_loss_mapping forms the same values."""
        flags = (bool(rm_dynamic), bool(dynamic))
        ent = self._held.get(id(viewpoint))
        if ent is not None and ent[0] is not viewpoint:     # the id was recycled by another object
            self.drop(ent[0])
            ent = None
        if ent is not None and flags in ent[3]:
            self._held[id(viewpoint)] = self._held.pop(id(viewpoint))           # (most recently used last)
            w = ent[3][flags]
            return ent[1], ent[2], w[0], w[1], w[2]
        gt_image, gt_depth, w_rgb, w_dep, alpha = slam_losses.mapping_loss_operands(config, viewpoint, device, rm_dynamic=rm_dynamic, dynamic=dynamic)
        if ent is None:
            ent = self._held[id(viewpoint)] = [viewpoint, gt_image, gt_depth, {}, self._nbytes(gt_image, gt_depth)]
            self._bytes += ent[4]
        else:
            self._held[id(viewpoint)] = self._held.pop(id(viewpoint))
        ent[3][flags] = (w_rgb, w_dep, alpha)
        extra = self._nbytes(w_rgb, w_dep)
        ent[4] += extra
        self._bytes += extra
        if self._budget is None:
            self._budget = device_store_budget(device, self.BUDGET_FRACTION)
        while self._bytes > self._budget and len(self._held) > 1:          # least recently used first, never the entry just returned
            oldest = next(iter(self._held))
            self._bytes -= self._held.pop(oldest)[4]
        return ent[1], ent[2], w_rgb, w_dep, alpha

    def drop(self, viewpoint=None):
        if viewpoint is None:
            self._held.clear()
            self._bytes = 0
        else:
            ent = self._held.pop(id(viewpoint), None)
            if ent is not None:
                self._bytes -= ent[4]

    def held_bytes(self):
        return self._bytes


class MappingGraph:
    """One run of plain static mapping iterations of ``backend`` over ``current_window`` (see the module docstring)."""

    def __init__(self, backend, current_window, viewpoint_stack, candidates, draws, iteration_count0):
        be = self.backend = backend
        g = be.gaussians
        self.current_window, self.window, self.candidates = list(current_window), list(viewpoint_stack), list(candidates)
        self.draws = [list(d) for d in draws]
        self.rows = len(self.draws)
        dev = self.device = self.window[0].device
        self.n_slots = min(N_INDEX_WORDS, len(self.candidates))
        proto = self.window[0]
        H, W = int(proto.image_height), int(proto.image_width)
        self.pixels = H * W
        cfg = be.config
        # ---- slots: a camera + ground truth / weights each ------------------------------------------------------------------------------
        self.slots, self.slot_ops = [], []
        for s in range(self.n_slots):
            cam = Camera(-1 - s, None, None, torch.eye(4), proto.projection_matrix, proto.fx, proto.fy, proto.cx, proto.cy, proto.FoVx, proto.FoVy,
                         H, W, 0.0, None, device=dev)
            self.slots.append(cam)
            self.slot_ops.append((torch.zeros((3, H, W), device=dev), torch.zeros((1, H, W), device=dev), torch.zeros((1, H, W), device=dev),
                                  torch.zeros((1, H, W), device=dev)))
        store = be.keyframe_operands
        self.window_ops = [store.get(cfg, v, dev, rm_dynamic=True, dynamic=False) for v in self.window]
        self.alpha = self.window_ops[0][4]
        # ---- the candidates' buffer addresses, one gsr_keyframe_entry per candidate, in device memory ----------------------------------------
        self._keep = []
        if self.n_slots:
            rows = []
            for v in self.candidates:
                ops = store.get(cfg, v, dev, rm_dynamic=True, dynamic=False)
                for t in ops[:4]:
                    if t.dtype != torch.float32 or not t.is_contiguous():
                        raise RuntimeError("MappingGraph: loss operands must be contiguous float32 tensors")
                self._keep.append(ops)
                rows.append([v.world_view_transform.data_ptr(), v.full_proj_transform.data_ptr(), v.camera_center.data_ptr(),
                             v.exposure_a.data_ptr(), v.exposure_b.data_ptr(), ops[0].data_ptr(), ops[1].data_ptr(), ops[2].data_ptr(), ops[3].data_ptr()])
            self.kf_table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
            self.slot_dst = (_lib.KeyframeEntry * self.n_slots)()
            for s, (cam, ops) in enumerate(zip(self.slots, self.slot_ops)):
                d = self.slot_dst[s]
                d.viewmatrix, d.full_proj, d.campos = cam._view.data_ptr(), cam._full.data_ptr(), cam._campos.data_ptr()
                d.exposure_a, d.exposure_b = cam.exposure_a.data_ptr(), cam.exposure_b.data_ptr()
                d.gt_image, d.gt_depth, d.w_rgb, d.w_depth = (t.data_ptr() for t in ops)
        # ---- the schedule: per iteration [index 0, index 1 | (step size, 1 / sqrt(bias correction 2)) per parameter tensor] -------------------
        opt = g.optimizer
        self.todo = opt.scheduled_segments()
        if self.todo is None:
            raise RuntimeError("MappingGraph: the optimizer state does not fit the fused scheduled step (run one eager iteration first)")
        self.row_words = N_INDEX_WORDS + 2 * len(self.todo)
        table = np.zeros((self.rows, self.row_words), dtype=np.uint32)
        coef = table[:, N_INDEX_WORDS:].view(np.float32)
        for j in range(self.rows):
            for s, c in enumerate(self.draws[j][:N_INDEX_WORDS]):
                table[j, s] = int(c)
            for k, (group, p) in enumerate(self.todo):
                lr = group["lr"]
                if j > 0 and group.get("name") == "xyz":          # update_learning_rate(iteration_count) ran after the previous step (GM:492-505)
                    lr = g.xyz_lr_at(iteration_count0 + j)
                coef[j, 2 * k], coef[j, 2 * k + 1] = opt.coefficients(lr, group["betas"], int(opt.state[p]["step"]) + j + 1)
        self.table = torch.from_numpy(table.view(np.int32)).pin_memory().to(dev, non_blocking=True)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.current = torch.zeros(self.row_words, dtype=torch.int32, device=dev)
        self.graph, self.pkgs, self.executed = None, None, 0

    # ---- the iteration (one code path: run directly, or captured and replayed) ---------------------------------------------------------------
    def iteration(self):
        be, g, dev = self.backend, self.backend.gaussians, self.device
        self.pkgs = None            # (the previous iteration's autograd graph dies here, not while the next one is being built)
        L = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(L.gsr_schedule_advance(self.counter.data_ptr(), self.table.data_ptr(), self.row_words, self.rows, self.current.data_ptr(),
                                              _lib.stream(dev)), "gsr_schedule_advance")
            if self.n_slots:
                _lib.check(L.gsr_slot_gather(self.n_slots, self.kf_table.data_ptr(), self.current.data_ptr(), self.slot_dst, self.pixels,
                                             _lib.stream(dev)), "gsr_slot_gather")
        views = self.window + self.slots
        ops = list(self.window_ops) + [o + (self.alpha,) for o in self.slot_ops]
        rendered = be._render_many(views, [(None, None, None)] * len(views))
        # (every term a root of ONE backward pass with the gradient 1 instead of a sum nobody reads: see DynamicMapping.iteration)
        terms = [slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], gt_image, gt_depth, w_rgb, w_dep, viewpoint.exposure_a, viewpoint.exposure_b,
                                              alpha, compute_value=False)
                 for viewpoint, pkg, (gt_image, gt_depth, w_rgb, w_dep, alpha) in zip(views, rendered, ops)]
        terms.append(be._isotropic_loss())
        torch.autograd.backward(terms, [be.unit_gradient(dev)] * len(terms))
        with torch.no_grad():
            for pkg in rendered:
                be._view_stats(pkg)
            g.optimizer.step_scheduled(self.todo, self.current[N_INDEX_WORDS:].data_ptr())
            g.optimizer.zero_grad(set_to_none=True)
            be._pose_updates(self.window, self.current_window)
            be._clear_camera_grads(self.slots)
        self.pkgs = rendered
        return rendered

    def warm_up(self, n):
        """`n` iterations executed directly, on a side stream (torch's capture protocol: autograd's stream bookkeeping must have seen the
        stream family the capture will use). They are iterations like any other: rows 0..n-1 of the schedule."""
        dev = self.device
        s = self.backend.graph_streams(dev)[0]
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(n):
                self.iteration()
                self.executed += 1
        torch.cuda.current_stream(dev).wait_stream(s)

    def capture(self):
        """Capture one iteration. Not through ``torch.cuda.graph``: its __enter__ runs gc.collect() and empties the allocator's cache -- tens of
        milliseconds per capture at SLAM sizes, and every mapping call captures anew (the map's tensors change with every keyframe). All
        captures of a back-end share ONE private memory pool; the previous graph is kept alive until this capture has begun, so the pool
        (and its blocks) survive from capture to capture."""
        be, dev = self.backend, self.device
        lazy_before = _C.set_option("lazy", 1)
        margin_before = _C.set_option("cap_margin_permille", CAPTURE_MARGIN_PERMILLE)
        tile_before = _C.set_option("cap_tile_margin_permille", CAPTURE_TILE_MARGIN_PERMILLE)
        s = be.graph_streams(dev)[1]
        s.wait_stream(torch.cuda.current_stream(dev))
        self.graph = torch.cuda.CUDAGraph()
        import time
        t0 = time.perf_counter()
        a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        try:
            with torch.cuda.stream(s):
                self.graph.capture_begin(pool=be.graph_pool(dev))
                t1 = time.perf_counter()
                try:
                    self.iteration()
                finally:
                    t2 = time.perf_counter()
                    self.graph.capture_end()
            t3 = time.perf_counter()
            self.capture_parts_ms = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0)
        finally:
            _C.set_option("lazy", lazy_before)       # the flags only matter while host code runs: replays never consult them
            _C.set_option("cap_margin_permille", margin_before)
            _C.set_option("cap_tile_margin_permille", tile_before)
        torch.cuda.current_stream(dev).wait_stream(s)
        be._graph_keepalive = self.graph             # (drops the previous run's graph: the pool now belongs to this one)
        return self

    def replay(self, n):
        for _ in range(n):
            self.graph.replay()
        self.executed += n

    # ---- undo (a replayed frame outgrew its buffers) -----------------------------------------------------------------------------------
    def snapshot(self):
        g = self.backend.gaussians
        tensors = [p for _, p in self.todo] + [g.optimizer.state[p][k] for _, p in self.todo for k in ("exp_avg", "exp_avg_sq")]
        tensors += [g.xyz_gradient_accum, g.denom, g.max_radii2D, self.counter]
        for v in self.window:
            tensors += [v._R, v._T, v._adam, v._converged, v.exposure_a, v.exposure_b, v.cam_rot_delta, v.cam_trans_delta]
        with torch.no_grad():
            self._snap = [(t, t.detach().clone()) for t in tensors]
        self._snap_executed = self.executed

    def restore(self):
        with torch.no_grad():
            for t, c in self._snap:
                t.detach().copy_(c)
            for v in self.window:
                v.refresh_matrices()
        self.executed = self._snap_executed

    def release(self):
        self.graph, self.pkgs, self._snap = None, None, None


class InitGraph:
    """A run of plain iterations of BackEnd.initialize_map (utils/slam_backend.py:237-296: one view, the mapping loss without exposure, Adam
    on the Gaussians; no camera step, no regulariser, a constant learning rate) as hipGraph replays: the single-view counterpart of
    MappingGraph. The schedule rows only carry Adam's coefficients. ``iteration()`` is one code path -- executed directly for the warm-up,
    captured once, replayed; iterations that densify or reset opacities stay in initialize_map's eager loop and end a run."""

    def __init__(self, backend, viewpoint, rm_dynamic, rows):
        be = self.backend = backend
        g = be.gaussians
        self.viewpoint, self.rows = viewpoint, int(rows)
        dev = self.device = viewpoint.device
        self.ops = be.keyframe_operands.get(be.config, viewpoint, dev, rm_dynamic=rm_dynamic, dynamic=False)
        opt = g.optimizer
        self.todo = opt.scheduled_segments()
        if self.todo is None:
            raise RuntimeError("InitGraph: the optimizer state does not fit the fused scheduled step (run one eager iteration first)")
        self.row_words = 2 * len(self.todo)
        table = np.zeros((self.rows, self.row_words), dtype=np.float32)
        for j in range(self.rows):
            for k, (group, p) in enumerate(self.todo):
                table[j, 2 * k], table[j, 2 * k + 1] = opt.coefficients(group["lr"], group["betas"], int(opt.state[p]["step"]) + j + 1)
        self.table = torch.from_numpy(table.view(np.int32)).pin_memory().to(dev, non_blocking=True)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.current = torch.zeros(self.row_words, dtype=torch.int32, device=dev)
        self.graph, self.pkg, self.executed = None, None, 0

    def iteration(self):
        be, g, dev, v = self.backend, self.backend.gaussians, self.device, self.viewpoint
        self.pkg = None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().gsr_schedule_advance(self.counter.data_ptr(), self.table.data_ptr(), self.row_words, self.rows, self.current.data_ptr(),
                                                       _lib.stream(dev)), "gsr_schedule_advance")
        pkg = be._render(v, (None, None, None))
        gt_image, gt_depth, w_rgb, w_dep, alpha = self.ops
        loss = slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], gt_image, gt_depth, w_rgb, w_dep, None, None, alpha, compute_value=False)
        loss.backward()
        with torch.no_grad():
            be._view_stats(pkg)
            g.optimizer.step_scheduled(self.todo, self.current.data_ptr())
            g.optimizer.zero_grad(set_to_none=True)
        # (the view's camera parameters keep accumulating their gradients, as in the eager loop and in the reference, whose initialize_map
        # never clears them: in place, into the tensors the warm-up iterations left -- which is why the snapshot covers them)
        self.pkg = pkg
        return pkg

    def warm_up(self, n):
        dev = self.device
        s = self.backend.graph_streams(dev)[0]
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(n):
                self.iteration()
                self.executed += 1
        torch.cuda.current_stream(dev).wait_stream(s)

    def capture(self):
        """(see MappingGraph.capture)"""
        be, dev = self.backend, self.device
        lazy_before = _C.set_option("lazy", 1)
        margin_before = _C.set_option("cap_margin_permille", INIT_CAPTURE_MARGIN_PERMILLE)
        tile_before = _C.set_option("cap_tile_margin_permille", CAPTURE_TILE_MARGIN_PERMILLE)
        s = be.graph_streams(dev)[1]
        s.wait_stream(torch.cuda.current_stream(dev))
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.stream(s):
                self.graph.capture_begin(pool=be.graph_pool(dev))
                try:
                    self.iteration()
                finally:
                    self.graph.capture_end()
        finally:
            _C.set_option("lazy", lazy_before)
            _C.set_option("cap_margin_permille", margin_before)
            _C.set_option("cap_tile_margin_permille", tile_before)
        torch.cuda.current_stream(dev).wait_stream(s)
        be._graph_keepalive = self.graph
        return self

    def replay(self, n):
        for _ in range(n):
            self.graph.replay()
        self.executed += n

    def snapshot(self):
        g = self.backend.gaussians
        tensors = [p for _, p in self.todo] + [g.optimizer.state[p][k] for _, p in self.todo for k in ("exp_avg", "exp_avg_sq")]
        tensors += [g.xyz_gradient_accum, g.denom, g.max_radii2D, self.counter]
        v = self.viewpoint
        tensors += [p.grad for p in (v.cam_rot_delta, v.cam_trans_delta, v.exposure_a, v.exposure_b) if p is not None and p.grad is not None]
        with torch.no_grad():
            self._snap = [(t, t.detach().clone()) for t in tensors]
        self._snap_executed = self.executed

    def restore(self):
        with torch.no_grad():
            for t, c in self._snap:
                t.detach().copy_(c)
        self.executed = self._snap_executed
