"""The DYNAMIC mapping call (utils/slam_backend.py:306-774, BackEnd.map with the control-node network) in a fixed layout, its runs of plain
iterations as hipGraph replays (second half; slam/mapping_graph.py is the static counterpart).

A dynamic iteration is: node network on ~100 time samples -> node blend at the views' times -> two regularisers on the random samples ->
multi-view render (window views + 2 random keyframes) -> fused mapping losses -> 2 flow renders per view + flow losses -> ONE backward ->
statistics -> camera steps -> Adam on the network (and, in the second half of the call, on the Gaussians). Eager: ~370 launches from Python
and autograd, 6.4 ms of wall time for 5.0 ms of device time (profiles/r04_mapping_iteration_launches_dynamic.json).

What the host decides per iteration in the reference -- and what therefore had to move into device memory before an iteration could be
recorded once and replayed:

  * which two RANDOM keyframes are rendered (:344-349): two persistent *slots* (camera + ground truth + loss weights, filled by
    gsr_slot_gather from a device table of the candidates' buffer addresses, slam/mapping_graph.py) -- plus, here, the slot's TIME, its
    flow PARTNER (the closest earlier keyframe, :299-304,:479-509: a second slot camera) and the pair's flow targets (six image planes,
    moved by a second gather through the same entry point), and the partner's time;
  * the random TIME SAMPLES of the two regularisers (:517-519,:646-648; 12 per window view, 10 per random keyframe): drawn on the host
    for a whole run up front -- the same torch.rand calls in the same order as iteration by iteration -- and stored as a row of the
    device-side schedule next to the slot indices and Adam's coefficients (gsr_schedule_advance copies row `counter` into the block the
    iteration reads);
  * the ORDER of the network's batch: the eager loop (BackEnd.map's body, kept for sharded runs and test doubles) names time samples by
    their host value, sorts and merges them; here the layout is fixed -- [the window views' and their partners' times, merged and sorted
    once per call | slot 0's time, its partner's | slot 1's ... | the samples, view by view in drawing order] -- and consumers address
    rows by position (ControlNodes.begin_iteration_indexed / regularisers_indexed). Same terms, summed in another order.

``iteration()`` is ONE code path: executed directly -- every iteration when ``Training.mapping_graph`` is off, the warm-up iterations of a
run, the iterations that densify or reset opacities (those replace the model's tensors: they end a run) -- or captured once and replayed.
Direct and replayed iterations are bit-identical (tests/test_hip_slam.py). The call's two halves (:350-355: the loss weights double on the
moving pixels in the first half; :337-338,:765-770: the Gaussians only step in the second) are different graphs: a run never crosses the
boundary. A replayed forward pass that outgrows its binning buffer is detected after the run (sticky overflow counters); the run is then
undone from a snapshot and repeated directly."""
import os
import time

import numpy as np
import torch

from diff_gaussian_rasterization import _C
import control_nodes
import slam_losses

from . import _lib
from .camera import Camera
from .deform_model import draw_loss_times, time_key
from .mapping_graph import N_INDEX_WORDS

# Head room of the binning buffers a captured iteration lays out (include/gs_rasterizer.h): while the node network trains alone (first half
# of a call) the moving object's Gaussians can swell and pile up for a few iterations -- measured: instance counts +10 %, the longest tile
# list x 2-4 -- and every overflow costs the whole run (undone, repeated directly).
CAPTURE_MARGIN_PERMILLE, CAPTURE_TILE_MARGIN_PERMILLE = (int(v) for v in os.environ.get("GSR_DYN_MARGINS", "2000,7000").split(","))
# the slots of an iteration's two random keyframes (and their flow renders) show another keyframe on every replay: the instance count at
# capture says little about the next draw (flow slot: 7 042 at capture, > 25 000 later = the 3x margin overrun, 99 iterations redone in the
# config #4 stand-in, 39 of 400 in the census) -- a floor in instances (~70 B each: 9 MB per view) instead of a ratio alone
CAPTURE_FLOOR_INSTANCES = int(os.environ.get("GSR_DYN_CAP_FLOOR", "131072"))
FLOW_CLIPS = os.environ.get("GSR_FLOW_CLIPS", "1") != "0"      # render the flow images only where the flow loss reads them (gsr_view.flow_clip)
FLOW_TARGET_BUDGET_FRACTION = 0.03  # ... and at most this share of the device memory free when the first target is formed
FLOW_TARGET_CACHE_MAX = 512       # keyframe pairs whose flow targets are kept (~7 MB each at 640x480); dropped ones are formed again on demand
WINDOW_SAMPLES, EXTRA_SAMPLES = (4, 8), (2, 8)        # (ARAP, elastic) time samples per window view / per random keyframe (:517-519,:646-648)


_region = torch.autograd.profiler.record_function          # names a part of the iteration for the profiler (tools/mapping_iteration_launches.py)


def eligible(be, views, candidates):
    """Can this map() call run in the fixed layout? (Else BackEnd.map's eager body takes it: sharded runs, test doubles, monocular input,
    fewer than three nodes, keyframes without a motion mask.)"""
    g, t = be.gaussians, be.config["Training"]
    if not t.get("dynamic_fixed_layout", True) or be.shard.active or be.loss_values or t.get("monocular", False):
        return False
    if not str(be.device).startswith("cuda") or g.deform is None or not g.deform_init or g.get_xyz.shape[0] == 0:
        return False
    if g.deform.deform.node_num < 3 or g.dyn_rows().shape[0] == 0 or not getattr(g.optimizer, "_fused_acc", False):
        return False
    return all(isinstance(v, Camera) and v.depth is not None and v.motion_mask is not None for v in list(views) + list(candidates))


class _Run:
    """`rows` consecutive iterations of one phase: their schedule in device memory."""
    __slots__ = ("i0", "rows", "dyn", "stepping", "with_flow", "flow_weight", "layout", "tables", "todo", "table", "row_words", "coef_lo", "samples_lo",
                 "n_rest", "current", "count0", "draws")


class DynamicMapping:
    """One BackEnd.map(current_window, iters, dynamic_network=True) call (see the module docstring)."""

    def __init__(self, backend, current_window, views, positions, candidates, iters, warm):
        be = self.be = backend
        self.g = be.gaussians
        self.nodes = self.g.deform.deform
        self.current_window, self.views, self.positions, self.candidates = list(current_window), list(views), list(positions), list(candidates)
        self.iters, self.warm = int(iters), int(warm)
        dev = self.device = self.views[0].device
        proto = self.views[0]
        self.H, self.W = int(proto.image_height), int(proto.image_width)
        self.pixels = self.H * self.W
        t = be.config["Training"]
        self.flow_weights = {True: float(t["flow_loss"]), False: float(t.get("flow_loss_fine", t["flow_loss"]))}
        self.has_flow_data = hasattr(be.dataset, "gt_flow")
        self.n_slots = min(N_INDEX_WORDS, len(self.candidates))
        any_flow = self.has_flow_data and max(self.flow_weights.values()) > 0
        mk = lambda uid: Camera(uid, None, None, torch.eye(4), proto.projection_matrix, proto.fx, proto.fy, proto.cx, proto.cy, proto.FoVx, proto.FoVy,
                                self.H, self.W, 0.0, None, device=dev)
        z = lambda c: torch.zeros((c, self.H, self.W), device=dev)
        self.slots = [mk(-1 - s) for s in range(self.n_slots)]
        self.slot_ops = [(z(3), z(1), z(1), z(1)) for _ in range(self.n_slots)]
        self.partner_slots = [mk(-101 - s) for s in range(self.n_slots)] if any_flow else []
        self.partner_flow = [z(6) for _ in range(self.n_slots)] if any_flow else []
        self.slot_dst = self._entries([(c, o) for c, o in zip(self.slots, self.slot_ops)])
        self.partner_dst = self._entries([(c, (f[0:3], f[3:4], f[4:5], f[5:6])) for c, f in zip(self.partner_slots, self.partner_flow)])
        self._zero6 = None
        self.slot_clips = torch.zeros((max(1, self.n_slots), 8), dtype=torch.int32, device=dev)       # per slot: its rectangle, its partner's
        self._layouts, self._tables, self._window_ops = {}, {}, {}
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.run, self.graph, self.pkgs = None, None, None
        self.stats = be.__dict__.setdefault("dynamic_graph_stats", {"runs": 0, "replays": 0, "direct": 0, "special": 0, "redone": 0, "failed": 0})

    @staticmethod
    def _entries(pairs):
        if not pairs:
            return None
        arr = (_lib.KeyframeEntry * len(pairs))()
        for d, (cam, ops) in zip(arr, pairs):
            d.viewmatrix, d.full_proj, d.campos = cam._view.data_ptr(), cam._full.data_ptr(), cam._campos.data_ptr()
            d.exposure_a, d.exposure_b = cam.exposure_a.data_ptr(), cam.exposure_b.data_ptr()
            d.gt_image, d.gt_depth, d.w_rgb, d.w_depth = (t.data_ptr() for t in ops)
        return arr

    # ---- per-call constants --------------------------------------------------------------------------------------------------------------
    def phase(self, i):
        """(the loss weights double on the moving pixels, :350-355; the Gaussians take part, :337-338,:765-770)"""
        return (i < self.iters / 2, i > self.warm)

    def with_flow(self, dyn):
        return self.has_flow_data and self.flow_weights[dyn] > 0

    def partner_of(self, v):
        """The keyframe the flow terms of `v` are formed with (:299-304,:479-509), or None."""
        closest = self.be.find_closest_keyframe(v.uid)
        if closest is None:
            return None
        other = self.be.viewpoints[closest]
        return other if (v.motion_mask is not None and other.motion_mask is not None) else None

    def flow6(self, v, other):
        """The constants of a keyframe pair's flow terms as ONE [6, H, W] tensor: planes 0-1 the flow v -> other on v's moving pixels, 2 their
        mask, 3 the mask of `other`, 4-5 the flow other -> v on other's moving pixels (:486-488,:503-505 mask both sides of the difference by
        ~motion_mask; with a 0 / 1 mask: the masked target minus the masked rendering). The plane order is the one gsr_slot_gather moves as
        (gt_image[3], gt_depth, w_rgb, w_depth). One entry per keyframe (its partner never changes), ~7 MB at 640x480, in a bounded store."""
        cache = self.be.__dict__.setdefault("_flow_targets6", {})
        hit = cache.get((v.uid, other.uid))
        if hit is None:
            ds = self.be.dataset
            m1 = (~v.motion_mask).to(torch.float32)[None]
            m2 = (~other.motion_mask).to(torch.float32)[None]
            back = ds.gt_flow(v.uid, other.uid)[0].permute(2, 0, 1) * m1
            fwd = ds.gt_flow(other.uid, v.uid)[0].permute(2, 0, 1) * m2
            hit = torch.cat([back, m1, m2, fwd], 0).to(self.device, torch.float32).contiguous()
            # bounded by bytes (a share of the memory free at the first use) as well as by count; oldest first; the tables of a running call
            # hold what they point at
            budget = self.be.__dict__.get("_flow_targets6_budget")
            if budget is None:
                from .mapping_graph import device_store_budget
                budget = self.be.__dict__["_flow_targets6_budget"] = device_store_budget(self.device, FLOW_TARGET_BUDGET_FRACTION)
            each = hit.numel() * hit.element_size()
            while cache and (len(cache) >= FLOW_TARGET_CACHE_MAX or (len(cache) + 1) * each > budget):
                cache.pop(next(iter(cache)))
            cache[(v.uid, other.uid)] = hit
        return hit

    def flow_clip(self, v):
        """The tile rectangle [x0, y0, x1, y1) (16-pixel tiles) around keyframe v's MOVING pixels -- all its flow loss reads of the flow image
        rendered from v -- as an int32 [4] device tensor (gsr_view.flow_clip); one host read per keyframe, kept."""
        cache = self.be.__dict__.setdefault("_flow_clips", {})
        hit = cache.get(v.uid)
        if hit is None:
            moving = ~v.motion_mask
            ys, xs = moving.any(dim=1).nonzero().flatten(), moving.any(dim=0).nonzero().flatten()
            rect = [0, 0, 0, 0] if ys.numel() == 0 else [int(xs[0]) // 16, int(ys[0]) // 16, int(xs[-1]) // 16 + 1, int(ys[-1]) // 16 + 1]
            while len(cache) >= 4 * FLOW_TARGET_CACHE_MAX:
                cache.pop(next(iter(cache)))
            hit = cache[v.uid] = torch.tensor(rect, dtype=torch.int32).to(self.device)
        return hit

    def layout(self, with_flow):
        hit = self._layouts.get(with_flow)
        if hit is None:
            partners = [self.partner_of(v) if with_flow else None for v in self.views]
            keys = sorted({time_key(v.time) for v in self.views} | {time_key(p.time) for p in partners if p is not None})
            index = {k: i for i, k in enumerate(keys)}
            epf = 2 if with_flow else 1                        # batch entries per slot: its time (and its partner's)
            n_w = len(keys)
            hit = self._layouts[with_flow] = {
                "partners": partners, "n_w": n_w, "epf": epf, "n_full": n_w + self.n_slots * epf,
                "view_idx": [index[time_key(v.time)] for v in self.views] + [n_w + e * epf for e in range(self.n_slots)],
                "partner_idx": [None if p is None else index[time_key(p.time)] for p in partners],
                "flow6": [None if p is None else self.flow6(v, p) for v, p in zip(self.views, partners)],
                "clips": [None if p is None else (self.flow_clip(v), self.flow_clip(p)) for v, p in zip(self.views, partners)],
                "wtimes": torch.tensor(keys, dtype=torch.float32).to(self.device)}
        return hit

    def tables(self, dyn, with_flow):
        """Per candidate keyframe, in device memory: the addresses gsr_slot_gather copies from (camera + loss operands; partner camera + flow
        targets) and the (own, partner) times."""
        hit = self._tables.get((dyn, with_flow))
        if hit is None:
            be, dev = self.be, self.device
            store, cfg = be.keyframe_operands, be.config
            rows, prow, times, keep, clip_rows = [], [], [], [], []
            for c in self.candidates:
                ops = store.get(cfg, c, dev, rm_dynamic=False, dynamic=dyn)
                for tns in ops[:4]:
                    if tns.dtype != torch.float32 or not tns.is_contiguous():
                        raise RuntimeError("DynamicMapping: loss operands must be contiguous float32 tensors")
                keep.append(ops)
                rows.append([c.world_view_transform.data_ptr(), c.full_proj_transform.data_ptr(), c.camera_center.data_ptr(), c.exposure_a.data_ptr(),
                             c.exposure_b.data_ptr(), ops[0].data_ptr(), ops[1].data_ptr(), ops[2].data_ptr(), ops[3].data_ptr()])
                if with_flow:
                    p = self.partner_of(c)
                    if p is None:               # no earlier keyframe: the candidate stands in for its partner with an all-zero target and mask --
                        p = c                   # both flow terms and their gradients are exactly zero
                        if self._zero6 is None:
                            self._zero6 = torch.zeros((6, self.H, self.W), device=dev)
                        f6 = self._zero6
                    else:
                        f6 = self.flow6(c, p)
                    keep.append(f6)
                    step = self.pixels * 4
                    prow.append([p.world_view_transform.data_ptr(), p.full_proj_transform.data_ptr(), p.camera_center.data_ptr(), p.exposure_a.data_ptr(),
                                 p.exposure_b.data_ptr(), f6.data_ptr(), f6.data_ptr() + 3 * step, f6.data_ptr() + 4 * step, f6.data_ptr() + 5 * step])
                    times.append([time_key(c.time), time_key(p.time)])
                    # (rows 2 c and 2 c + 1: the rectangles the two flow images of the pair are read in; the stand-in pair reads nothing)
                    clip_rows += [self.flow_clip(c), self.flow_clip(p)] if p is not c else [torch.zeros(4, dtype=torch.int32, device=dev)] * 2
                else:
                    times.append([time_key(c.time)])
            up = lambda a, dt: torch.tensor(a, dtype=dt).pin_memory().to(dev, non_blocking=True)
            hit = self._tables[(dyn, with_flow)] = {
                "keep": keep, "kf": up(rows, torch.int64) if rows else None, "partner": up(prow, torch.int64) if prow else None,
                "times": up(times, torch.float32) if times else None, "clips": torch.stack(clip_rows).reshape(len(self.candidates), 8) if clip_rows else None}
        return hit

    # ---- a run's schedule -------------------------------------------------------------------------------------------------------------------
    def make_run(self, i0, rows, phase):
        """Draw what the iterations i0 .. i0 + rows - 1 draw, in their order (:344-349 torch.randperm for the random keyframes, then per view
        the regularisers' torch.rand calls, deform_model.draw_loss_times), and lay it out as the run's device-side schedule."""
        be, g = self.be, self.g
        r = _Run()
        r.i0, r.rows, (r.dyn, r.stepping) = i0, rows, phase
        r.with_flow = self.with_flow(r.dyn)
        r.flow_weight = self.flow_weights[r.dyn]
        r.layout, r.tables = self.layout(r.with_flow), self.tables(r.dyn, r.with_flow)
        t = be.config["Training"]
        ti = g.time_interval
        arap_delta = float(t.get("delta", 5)) * ti                                    # :325,:518
        nv = len(self.views)
        r.draws = []
        for _ in range(rows):
            extra_idx = [int(c) for c in torch.randperm(len(self.candidates))[:2]]
            samples = []
            for k, v in enumerate(self.views + [self.candidates[c] for c in extra_idx]):
                window = k < nv
                plan = draw_loss_times(v.time, arap_delta if window else 5 * ti, WINDOW_SAMPLES[0] if window else EXTRA_SAMPLES[0], 5 * ti)
                samples += [time_key(x) for x in plan["arap"] + plan["elastic"]]
            r.draws.append((extra_idx, samples))
        r.n_rest = nv * sum(WINDOW_SAMPLES) + self.n_slots * sum(EXTRA_SAMPLES)
        opt = g.optimizer
        r.todo = opt.scheduled_segments() if r.stepping else None
        n_coef = 2 * len(r.todo) if r.todo else 0
        r.coef_lo, r.samples_lo = N_INDEX_WORDS, N_INDEX_WORDS + n_coef
        r.row_words = r.samples_lo + r.n_rest
        r.count0 = be.iteration_count
        table = np.zeros((rows, r.row_words), dtype=np.uint32)
        fl = table.view(np.float32)
        for j, (extra_idx, samples) in enumerate(r.draws):
            for s, c in enumerate(extra_idx[:N_INDEX_WORDS]):
                table[j, s] = c
            for k, (group, p) in enumerate(r.todo or ()):
                lr = group["lr"]
                if j > 0 and group.get("name") == "xyz":          # update_learning_rate(iteration_count) ran after the previous step (GM:492-505)
                    lr = g.xyz_lr_at(r.count0 + j)
                fl[j, r.coef_lo + 2 * k], fl[j, r.coef_lo + 2 * k + 1] = opt.coefficients(lr, group["betas"], int(opt.state[p]["step"]) + j + 1)
            fl[j, r.samples_lo:] = np.asarray(samples, dtype=np.float32)
        r.table = torch.from_numpy(table.view(np.int32)).pin_memory().to(self.device, non_blocking=True)
        r.current = torch.zeros(r.row_words, dtype=torch.int32, device=self.device)
        self.counter.zero_()
        self.run, self.graph = r, None
        return r

    # ---- the iteration (one code path: run directly, or captured and replayed) ---------------------------------------------------------------
    def iteration(self, special=None):
        """`special` (direct execution only): {"last": bool} for an iteration that may densify / reset opacities -- it decides that on the
        host like the eager loop and takes a plain optimizer step (the model's tensors may have been replaced)."""
        be, g, nodes, r, dev = self.be, self.g, self.nodes, self.run, self.device
        lay, tab = r.layout, r.tables
        self.pkgs = None            # (the previous iteration's autograd graph dies here, not while the next one is being built)
        L = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(L.gsr_schedule_advance(self.counter.data_ptr(), r.table.data_ptr(), r.row_words, r.rows, r.current.data_ptr(), _lib.stream(dev)),
                       "gsr_schedule_advance")
            if self.n_slots:
                _lib.check(L.gsr_slot_gather(self.n_slots, tab["kf"].data_ptr(), r.current.data_ptr(), self.slot_dst, self.pixels, _lib.stream(dev)),
                           "gsr_slot_gather")
                if r.with_flow:
                    _lib.check(L.gsr_slot_gather(self.n_slots, tab["partner"].data_ptr(), r.current.data_ptr(), self.partner_dst, self.pixels,
                                                 _lib.stream(dev)), "gsr_slot_gather")
        # ---- the iteration's time samples, in the fixed layout ------------------------------------------------------------------------------
        # (the _region ranges only name the iteration's parts for tools/mapping_iteration_launches.py; no effect on the work)
        with _region("gsr.network"):
            parts = [lay["wtimes"]]
            if self.n_slots:
                parts.append(tab["times"].index_select(0, r.current[:self.n_slots].long()).reshape(-1))
            parts.append(r.current.view(torch.float32)[r.samples_lo:r.samples_lo + r.n_rest])
            it = nodes.begin_iteration_indexed(torch.cat(parts), lay["n_full"], blend=(g.get_dygs_xyz.detach(), g.motion_mask))
        nv, ne = len(self.views), self.n_slots
        with _region("gsr.regularisers"):
            loss_network = nodes.regularisers_indexed(it, nv, ne, be._regulariser_weights(nv, ne), WINDOW_SAMPLES, EXTRA_SAMPLES)
        views = self.views + self.slots
        # Who reads which row of the warp's output: every view's render its own sample (d_xyz, d_scaling, d_rotation); a flow pair (view sample
        # i, partner sample j) reads i whole + j's d_xyz for "this keyframe -> the earlier one", and j whole + i's d_xyz for the way back. All
        # readers are outputs of ONE node (control_nodes.fan_out) whose backward adds a row's gradients in one launch for all rows.
        plan = []

        def reader(i, whole):
            pos = {"x": len(plan)}
            plan.append((0, i))
            if whole:
                pos["s"], pos["r"] = len(plan), len(plan) + 1
                plan.extend([(2, i), (1, i)])
            return pos

        render_readers = [reader(i, True) for i in lay["view_idx"]]
        flow_readers = {}
        if r.with_flow:
            for k in range(len(views)):
                if k < nv and lay["partners"][k] is None:
                    continue
                i = lay["view_idx"][k]
                j = lay["partner_idx"][k] if k < nv else i + 1
                flow_readers[k] = (reader(i, True), reader(j, False), reader(j, True), reader(i, False))
        fan = control_nodes.fan_out(it["blended_stacked"], plan)
        deltas = [(fan[p["x"]], fan[p["s"]], fan[p["r"]]) for p in render_readers]
        cfg = be.config
        ops = self._window_ops.get(r.dyn)          # (held by the call: a captured iteration points at these buffers)
        if ops is None:
            ops = self._window_ops[r.dyn] = [be.keyframe_operands.get(cfg, v, dev, rm_dynamic=False, dynamic=r.dyn) for v in self.views]
        ops = ops + [o + (ops[0][4],) for o in self.slot_ops]
        with _region("gsr.render"):
            rendered = be._render_many(views, deltas)
        # The iteration's loss is a sum of terms whose VALUE nobody reads (the fused losses leave it uninitialised): instead of adding them up --
        # a launch per term -- every term is a root of ONE backward pass with the gradient 1 (be.unit_gradient): the same gradients, exactly
        terms = [loss_network]
        with _region("gsr.losses"):
            for v, pkg, (gt_image, gt_depth, w_rgb, w_dep, alpha) in zip(views, rendered, ops):
                terms.append(slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], gt_image, gt_depth, w_rgb, w_dep, v.exposure_a,
                                                          v.exposure_b, alpha, compute_value=False))
        if r.with_flow:
            from gaussian_renderer import render_flow_views
            requests, pairs, clips = [], [], []
            if self.n_slots and tab["clips"] is not None:
                torch.index_select(tab["clips"], 0, r.current[:self.n_slots].long(), out=self.slot_clips[:self.n_slots])
            for k, v in enumerate(views):
                if k < nv:
                    other, f6 = lay["partners"][k], lay["flow6"][k]
                    if other is None:
                        continue
                    clips += list(lay["clips"][k])
                else:
                    other, f6 = self.partner_slots[k - nv], self.partner_flow[k - nv]
                    clips += [self.slot_clips[k - nv, 0:4], self.slot_clips[k - nv, 4:8]]
                a1, b1, a2, b2 = flow_readers[k]
                requests += [(v, other, fan[a1["x"]], fan[b1["x"]], fan[a1["r"]], fan[a1["s"]]),      # this keyframe -> the earlier one,
                             (other, v, fan[a2["x"]], fan[b2["x"]], fan[a2["r"]], fan[a2["s"]])]      # and back
                pairs.append((f6[0:2], f6[2:3], f6[4:6], f6[3:4]))
            if requests:
                with _region("gsr.flow_render"):
                    flows = render_flow_views(g, requests, clips=clips if FLOW_CLIPS else None)
                if FLOW_CLIPS and be.config["Training"].get("flow_clip_check") and not torch.cuda.is_current_stream_capturing():
                    self._check_flow_clips(requests, pairs, clips, flows)          # TEST facility: the same terms without the clips
                with _region("gsr.flow_losses"):
                    for k, (t_back, m1, t_fwd, m2) in enumerate(pairs):
                        terms.append(slam_losses.masked_l1(r.flow_weight, [(flows[2 * k]["render"], t_back, m1), (flows[2 * k + 1]["render"], t_fwd, m2)],
                                                           channels=2, compute_value=False))          # (nobody reads the value: eligible() excludes loss_values)
        with _region("gsr.isotropic"):
            terms.append(be._isotropic_loss())
        torch.autograd.backward(terms, [be.unit_gradient(dev)] * len(terms))
        with _region("gsr.end_iteration"):
            nodes.end_iteration()
        split = False
        with torch.no_grad(), _region("gsr.updates"):
            if special is not None and special.get("last"):           # (before a densification changes the row count, like the eager loop)
                be._publish_visibility(self.current_window, {k: rendered[k]["n_touched"] for k in range(nv)}, n_views=nv)
            for pkg in rendered:
                be._view_stats(pkg)
            if special is not None and r.stepping:
                count = be.iteration_count
                update_gaussian = count % be.gaussian_update_every == be.gaussian_update_offset
                if update_gaussian:
                    g.densify_and_prune(be.opt_params.densify_grad_threshold, be.gaussian_th, be.gaussian_extent, be.size_threshold)
                    split = True
                if count % be.gaussian_reset == 0 and not update_gaussian:
                    be._reset_opacity_of_unseen(rendered)
                    split = True
            be._pose_updates(self.views, self.current_window, positions=self.positions)
            be._clear_camera_grads(self.slots + self.partner_slots)
            g.deform.optimizer.step()
            g.deform.optimizer.zero_grad(set_to_none=True)
            if r.stepping:                                                    # :765-770
                if special is None:
                    g.optimizer.step_scheduled(r.todo, r.current[r.coef_lo:].data_ptr())
                else:
                    g.optimizer.step()
                    g.update_learning_rate(be.iteration_count)
            g.optimizer.zero_grad(set_to_none=True)
        self.pkgs = rendered
        return split

    def _check_flow_clips(self, requests, pairs, clips, flows):
        """TEST facility (Training.flow_clip_check): render the iteration's flow images again WITHOUT the clips and record, in
        backend.flow_clip_checks, whether the masked images are equal, how far the gradients of the flow loss are apart (relative to their
        largest entry) and how many Gaussians each variant rasterized."""
        from gaussian_renderer import render_flow_views
        be, g, r = self.be, self.g, self.run

        def grads_of(cl):
            fl = render_flow_views(g, requests, clips=cl)
            loss = 0.0
            for k, (t_back, m1, t_fwd, m2) in enumerate(pairs):
                loss = loss + slam_losses.masked_l1(r.flow_weight, [(fl[2 * k]["render"], t_back, m1), (fl[2 * k + 1]["render"], t_fwd, m2)], channels=2)
            leaves = [g._xyz] + [t for rq in requests for t in rq[2:]]
            out = torch.autograd.grad(loss, leaves, allow_unused=True)
            g.optimizer.zero_grad(set_to_none=True)
            return fl, out
        _, ga = grads_of(clips)
        ref, gb = grads_of(None)
        rel = 0.0
        for x, y in zip(ga, gb):
            if x is not None and y is not None and float(y.abs().max()) > 0:
                rel = max(rel, float((x - y).abs().max() / y.abs().max()))
        same = True
        with torch.no_grad():
            for k, (t_back, m1, t_fwd, m2) in enumerate(pairs):
                for j, m in ((2 * k, m1), (2 * k + 1, m2)):
                    same = same and torch.equal(flows[j]["render"][:2] * m, ref[j]["render"][:2] * m)
            drawn = (sum(int((f["radii"] > 0).sum()) for f in flows), sum(int((f["radii"] > 0).sum()) for f in ref))
        be.__dict__.setdefault("flow_clip_checks", []).append({"masked_images_equal": same, "gradient_rel_diff": rel, "gaussians_drawn": drawn})

    # ---- executing a run ---------------------------------------------------------------------------------------------------------------
    def finish(self, n):
        """Host-side state after `n` plain iterations of the current run: what n eager iterations would have left."""
        be, g, r = self.be, self.g, self.run
        be.last_sent += n
        if r.stepping and n:
            be.iteration_count += n
            g.optimizer.advance_steps(r.todo, n)
            g.update_learning_rate(be.iteration_count)

    def direct(self, n):
        for _ in range(n):
            self.iteration()
        self.stats["direct"] += n

    def warm_up(self, n):
        """`n` iterations executed directly on a side stream (torch's capture protocol: autograd's stream bookkeeping must have seen the
        stream family the capture will use)."""
        dev = self.device
        s = self.be.graph_streams(dev)[0]
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(n):
                self.iteration()
        torch.cuda.current_stream(dev).wait_stream(s)

    def capture(self):
        """(raw capture API on a persistent side stream and ONE private memory pool per back-end: see MappingGraph.capture)"""
        be, dev = self.be, self.device
        lazy_before = _C.set_option("lazy", 1)
        margin_before = _C.set_option("cap_margin_permille", CAPTURE_MARGIN_PERMILLE)
        tile_before = _C.set_option("cap_tile_margin_permille", CAPTURE_TILE_MARGIN_PERMILLE)
        floor_before = _C.set_option("cap_floor", CAPTURE_FLOOR_INSTANCES)
        # TEST facility (Training.graph_test_shrink_permille): lay the captured buffers out too small, so that replays overflow
        shrink_before = _C.set_option("cap_test_shrink_permille", int(be.config["Training"].get("graph_test_shrink_permille", 0)))
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
            _C.set_option("lazy", lazy_before)       # the flags only matter while host code runs: replays never consult them
            _C.set_option("cap_margin_permille", margin_before)
            _C.set_option("cap_tile_margin_permille", tile_before)
            _C.set_option("cap_floor", floor_before)
            _C.set_option("cap_test_shrink_permille", shrink_before)
        torch.cuda.current_stream(dev).wait_stream(s)
        be._graph_keepalive = self.graph             # (drops the previous run's graph: the pool now belongs to this one)

    def snapshot(self):
        g, r = self.g, self.run
        tensors = [p for grp in g.optimizer.param_groups for p in grp["params"]]
        if r.todo:
            tensors += [g.optimizer.state[p][k] for _, p in r.todo for k in ("exp_avg", "exp_avg_sq")]
        net = g.deform.optimizer
        for grp in net.param_groups:
            for p in grp["params"]:
                tensors.append(p)
                tensors += [v for v in net.state.get(p, {}).values() if torch.is_tensor(v)]
        tensors += [g.xyz_gradient_accum, g.denom, g.max_radii2D, self.counter]
        for v in self.views:
            tensors += [v._R, v._T, v._adam, v._converged, v.exposure_a, v.exposure_b, v.cam_rot_delta, v.cam_trans_delta]
        with torch.no_grad():
            self._snap = [(t, t.detach().clone()) for t in tensors]

    def restore(self):
        with torch.no_grad():
            for t, c in self._snap:
                t.detach().copy_(c)
            for v in self.views:
                v.refresh_matrices()

    def plain_run(self, rows):
        """A run of plain iterations: replays of one captured iteration when the run is long enough and graphs are on, else direct."""
        be, r = self.be, self.run
        use_graph = (be._graphs_enabled() and rows >= be.graph_min_run and not getattr(be, "_dynamic_graph_broken", False)
                     and (not r.stepping or r.todo is not None))
        if not use_graph:
            self.direct(rows)
            self.finish(rows)
            return
        warm = min(be.dynamic_graph_warmup, rows)
        self.warm_up(warm)
        self.stats["direct"] += warm
        overflow0 = _C.forward_status_views()
        slots_before = _C.debug_view_slots(100)
        self.snapshot()
        try:
            t0 = time.perf_counter()
            self.capture()
            self.stats["capture_ms"] = self.stats.get("capture_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        except Exception as e:        # a capture that fails leaves the iterations done so far valid: finish the run directly, stop capturing
            be._dynamic_graph_broken = True
            torch.cuda.synchronize(self.device)
            self.restore()
            self.g.optimizer.zero_grad(set_to_none=True)
            self.g.deform.optimizer.zero_grad(set_to_none=True)
            be._clear_camera_grads(self.views + self.slots + self.partner_slots)
            self.stats["failed"] += 1
            self.stats["last_error"] = f"{type(e).__name__}: {e}"
            if be.config["Training"].get("mapping_graph") == "strict":
                raise
            self.graph = None
            self.direct(rows - warm)
            self.finish(rows)
            return
        for _ in range(rows - warm):
            self.graph.replay()
        torch.cuda.current_stream(self.device).synchronize()
        if _C.forward_status_views() != overflow0:          # a replayed view outgrew its binning buffer: undo the replays, redo them directly
            # (which slot outgrew what, kept in the statistics: slots whose overflow counter moved, with the estimates the layout started from)
            slots_after = _C.debug_view_slots(100)
            moved = [dict(slot=k, **{n: a[n] for n in ("R_alloc", "longest_tile", "estimate_R_alloc", "estimate_longest_tile")},
                          captured_estimate_R=b["estimate_R_alloc"], captured_estimate_tile=b["estimate_longest_tile"])
                     for k, (a, b) in enumerate(zip(slots_after, slots_before)) if a["overflows"] != b["overflows"]]
            self.stats.setdefault("overflow_causes", []).append({"rows": int(rows - warm), "gaussians": int(self.g.get_xyz.shape[0]), "slots": moved[:6]})
            self.restore()
            self.stats["redone"] += rows - warm
            self.graph = None
            self.direct(rows - warm)
        else:
            self.stats["replays"] += rows - warm
            self.stats["runs"] += 1
        self.finish(rows)

    def execute(self):
        """All iterations of the call. Returns gaussian_split of the last one (:336,:745)."""
        be, g = self.be, self.g
        i, split = 0, False
        nv = len(self.views)
        while i < self.iters:
            ph = self.phase(i)
            stepping = ph[1]

            def special_at(k, count):          # does iteration k (whose iteration_count would be `count`) densify, reset, or lack Adam moments?
                return stepping and (count % be.gaussian_update_every == be.gaussian_update_offset or count % be.gaussian_reset == 0)

            count = be.iteration_count + (1 if stepping else 0)
            needs_state = stepping and g.optimizer.scheduled_segments() is None
            if special_at(i, count) or needs_state:
                if stepping:
                    be.iteration_count += 1                                      # :337-338
                be.last_sent += 1
                self.make_run(i, 1, ph)
                split = self.iteration(special={"last": i == self.iters - 1})
                self.stats["special"] += 1
                i += 1
                continue
            rows = 0
            while i + rows < self.iters and self.phase(i + rows) == ph and not special_at(i + rows, count + rows):
                rows += 1
            self.make_run(i, rows, ph)
            self.plain_run(rows)
            split = False
            i += rows
            if i == self.iters:
                with torch.no_grad():
                    be._publish_visibility(self.current_window, {k: self.pkgs[k]["n_touched"] for k in range(nv)}, n_views=nv)
        self.run, self.graph, self.pkgs, self._snap = None, None, None, None
        return split


class NetworkInit:
    """BackEnd.initialize_network's loop (utils/slam_backend.py:160-234: the node network fitted on ONE view -- network, blend, render, the
    mapping loss without exposure, backward, Adam on the network) in the indexed layout (one time sample, every head), its iterations after the
    first -- which may densify -- as hipGraph replays. ``iteration()`` is one code path, executed directly or captured and replayed."""

    def __init__(self, backend, viewpoint):
        be = self.be = backend
        self.g, self.viewpoint = be.gaussians, viewpoint
        self.nodes = self.g.deform.deform
        dev = self.device = viewpoint.device
        self.ops = be.keyframe_operands.get(be.config, viewpoint, dev, rm_dynamic=False, dynamic=False)
        self.time = torch.tensor([time_key(viewpoint.time)], dtype=torch.float32).to(dev)
        self.graph, self.pkg = None, None
        self.stats = be.__dict__.setdefault("network_init_graph_stats", {"runs": 0, "replays": 0, "direct": 0, "redone": 0, "failed": 0})

    @staticmethod
    def eligible(be, viewpoint, update_gaussians):
        g, t = be.gaussians, be.config["Training"]
        return (t.get("dynamic_fixed_layout", True) and not update_gaussians and not be.shard.active and not be.loss_values and not t.get("monocular", False)
                and str(be.device).startswith("cuda") and isinstance(viewpoint, Camera) and viewpoint.depth is not None and g.deform_init
                and g.deform.deform.node_num >= 1 and g.dyn_rows().shape[0] > 0 and getattr(g.optimizer, "_fused_acc", False))

    def iteration(self, densify=False):
        be, g, nodes, v = self.be, self.g, self.nodes, self.viewpoint
        self.pkg = None
        it = nodes.begin_iteration_indexed(self.time, 1, blend=(g.get_dygs_xyz.detach(), g.motion_mask))
        rows = it["blended"]
        pkg = be._render(v, (rows[0][0], rows[2][0], rows[1][0]))
        gt_image, gt_depth, w_rgb, w_dep, alpha = self.ops
        slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], gt_image, gt_depth, w_rgb, w_dep, None, None, alpha, compute_value=False).backward()
        nodes.end_iteration()
        with torch.no_grad():
            be._view_stats(pkg)
            if densify:                                                            # :209-215 (iteration 0 with the shipped schedule)
                g.densify_and_prune(be.opt_params.densify_grad_threshold, be.init_gaussian_th, be.init_gaussian_extent, None)
            g.deform.optimizer.step()
            g.deform.optimizer.zero_grad(set_to_none=True)
            g.optimizer.zero_grad(set_to_none=True)
        # (the view's camera parameters keep accumulating their gradients, as in the reference: see mapping_graph.InitGraph)
        self.pkg = pkg
        return pkg

    def _snapshot(self):
        g, v = self.g, self.viewpoint
        net = g.deform.optimizer
        tensors = []
        for grp in net.param_groups:
            for p in grp["params"]:
                tensors.append(p)
                tensors += [s for s in net.state.get(p, {}).values() if torch.is_tensor(s)]
        tensors += [g.xyz_gradient_accum, g.denom, g.max_radii2D]
        tensors += [p.grad for p in (v.cam_rot_delta, v.cam_trans_delta, v.exposure_a, v.exposure_b) if p is not None and p.grad is not None]
        with torch.no_grad():
            return [(t, t.detach().clone()) for t in tensors]

    def run(self, iterations):
        """All iterations of the loop; returns the last render package."""
        be, dev = self.be, self.device
        i = 0
        while i < iterations:
            if i % be.init_gaussian_update == 0:
                self.iteration(densify=True)
                self.stats["direct"] += 1
                i += 1
                continue
            rows = 0
            while i + rows < iterations and (i + rows) % be.init_gaussian_update != 0:
                rows += 1
            if not (be._graphs_enabled() and rows >= be.graph_min_run and not getattr(be, "_network_init_graph_broken", False)):
                for _ in range(rows):
                    self.iteration()
                self.stats["direct"] += rows
                i += rows
                continue
            warm = min(be.dynamic_graph_warmup, rows)
            s0, s1 = be.graph_streams(dev)
            s0.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s0):
                for _ in range(warm):
                    self.iteration()
            torch.cuda.current_stream(dev).wait_stream(s0)
            self.stats["direct"] += warm
            overflow0 = _C.forward_status()[0]
            snap = self._snapshot()
            ok = True
            lazy_before = _C.set_option("lazy", 1)
            margin_before = _C.set_option("cap_margin_permille", CAPTURE_MARGIN_PERMILLE)
            tile_before = _C.set_option("cap_tile_margin_permille", CAPTURE_TILE_MARGIN_PERMILLE)
            floor_before = _C.set_option("cap_floor", CAPTURE_FLOOR_INSTANCES)
            try:
                s1.wait_stream(torch.cuda.current_stream(dev))
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(s1):
                    self.graph.capture_begin(pool=be.graph_pool(dev))
                    try:
                        self.iteration()
                    finally:
                        self.graph.capture_end()
                torch.cuda.current_stream(dev).wait_stream(s1)
                be._graph_keepalive = self.graph
            except Exception as e:
                be._network_init_graph_broken = True
                torch.cuda.synchronize(dev)
                self.stats["failed"] += 1
                self.stats["last_error"] = f"{type(e).__name__}: {e}"
                if be.config["Training"].get("mapping_graph") == "strict":
                    raise
                ok = False
            finally:
                _C.set_option("lazy", lazy_before)
                _C.set_option("cap_margin_permille", margin_before)
                _C.set_option("cap_tile_margin_permille", tile_before)
                _C.set_option("cap_floor", floor_before)
            if ok:
                for _ in range(rows - warm):
                    self.graph.replay()
                torch.cuda.current_stream(dev).synchronize()
                ok = _C.forward_status()[0] == overflow0
            if ok:
                self.stats["replays"] += rows - warm
                self.stats["runs"] += 1
            else:                               # undo whatever the capture / the replays did, repeat directly
                with torch.no_grad():
                    for t, c in snap:
                        t.detach().copy_(c)
                self.g.optimizer.zero_grad(set_to_none=True)
                self.g.deform.optimizer.zero_grad(set_to_none=True)
                self.graph = None
                self.stats["redone"] += rows - warm
                for _ in range(rows - warm):
                    self.iteration()
            i += rows
        pkg, self.graph = self.pkg, None
        return pkg
