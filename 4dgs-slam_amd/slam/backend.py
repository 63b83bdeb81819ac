"""Mapping back-end -- the counterpart of the reference's ``utils/slam_backend.py`` BackEnd, single process: add_next_kf (:107-128),
initialize_map (:237-296), initialize_network (:160-234), map_static (:1013-1224), map (:306-774, the dynamic branch with the
control-node warp), color_refinement (:777-862) and the message handlers of run() (:879-1010) as direct calls.

Per mapping iteration: every window keyframe is rendered through the fused prologue, the mapping loss is the fused weighted L1, the
densification statistics of a view are one launch, pose / exposure updates are one launch per camera, the six Gaussian groups are
stepped by FusedAdam, and densify_and_prune rebuilds the model with one launch (slam/gaussian_model.py).

Multi-GPU (north_star: "the mapping backend's per-keyframe render+backward is sharded across GPUs"): with a process group
up, every rank runs this class on a replica of the map and the views of an iteration -- window keyframes, then the random ones -- are
rendered, back-propagated and pose-stepped by rank ``index % world`` only (mapping_shard.ViewShard). Before the optimizers step, the
Gaussian gradients (the optimizer's flat bucket: the backward kernels already summed the rank's views into it) and the node network's are
all-reduced; the densification statistics, the visibility union of the opacity reset and the window keyframes' visibility rows are
reduced only on the iterations that use them; the window cameras' poses / exposures travel owner -> everyone once per iteration (14
floats each). The view-independent terms (isotropic-scale regulariser, ARAP / elastic node regularisers) are added on rank 0. All ranks
must be seeded alike: they take the same random draws (extra keyframes, time samples, split noise)."""
import os
import random
import time

import torch

from gaussian_renderer import render, render_views
import mapping_shard
import slam_losses

from .deform_model import draw_loss_times


class _IsotropicLoss(torch.autograd.Function):
    """10 * mean |s - mean_k s| of the [P, 3] scales s = exp(raw) (utils/slam_backend.py:653-655) through gsr_isotropic_loss_forward /
    _backward (include/slam_map.h): the dozen launches of the tensor expression and its backward as three; sums in a fixed order."""

    @staticmethod
    def forward(ctx, raw):
        from . import _lib
        raw_c = raw.detach().contiguous()
        P = int(raw_c.shape[0])
        L = _lib.lib()
        loss = torch.empty((), dtype=torch.float32, device=raw.device)
        ws = torch.empty((int(L.gsr_isotropic_loss_workspace_size(P)),), dtype=torch.uint8, device=raw.device)
        with torch.cuda.device(raw.device):
            _lib.check(L.gsr_isotropic_loss_forward(P, raw_c.data_ptr(), loss.data_ptr(), ws.data_ptr(), _lib.stream(raw.device)), "gsr_isotropic_loss_forward")
        ctx.save_for_backward(raw_c)
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        (raw_c,) = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        out = torch.empty_like(raw_c)
        with torch.cuda.device(raw_c.device):
            _lib.check(_lib.lib().gsr_isotropic_loss_backward(int(raw_c.shape[0]), raw_c.data_ptr(), g.data_ptr(), out.data_ptr(), _lib.stream(raw_c.device)),
                       "gsr_isotropic_loss_backward")
        return out


class BackEnd:
    def __init__(self, config):
        self.config = config
        self.gaussians = None
        self.pipeline_params = None
        self.opt_params = None
        self.background = None
        self.cameras_extent = None
        self.dataset = None
        self.device = "cuda"
        self.monocular = config["Training"].get("monocular", False)
        self.iteration_count = 0
        self.last_sent = 0
        self.occ_aware_visibility = {}
        self.viewpoints = {}
        self.current_window = []
        self.initialized = not self.monocular
        self.dynamic_model = config["model_params"]["dynamic_model"]
        self.dystart = 0
        self.pose_lr_scale = 0.5
        self.frames_to_optimize = config["Training"]["pose_window"]
        self.log = []
        self._view_shard = None
        self._flow_targets = {}          # (keyframe, earlier keyframe) -> masked flow targets of _flow_loss

    @property
    def shard(self):
        """View ownership + collectives of the current process group (a no-op object in a single process)."""
        if self._view_shard is None:
            self._view_shard = mapping_shard.ViewShard()
        return self._view_shard

    def attach_process_group(self, group=None):
        self._view_shard = mapping_shard.ViewShard(group)
        return self._view_shard

    def set_hyperparams(self):
        # Training.loss_values (default False): the mapping loops only back-propagate their losses, so the value's two launches per view are
        # skipped and -- in the static loop -- every view is back-propagated on its own (same gradient sum, accumulated in view order)
        self.loss_values = bool(self.config["Training"].get("loss_values", False))
        """:81-105."""
        t = self.config["Training"]
        self.save_results, self.save_dir = self.config["Results"].get("save_results", False), self.config["Results"].get("save_dir")
        self.init_itr_num, self.init_gaussian_update = t["init_itr_num"], t["init_gaussian_update"]
        self.init_gaussian_reset, self.init_gaussian_th = t["init_gaussian_reset"], t["init_gaussian_th"]
        self.init_gaussian_extent = self.cameras_extent * t["init_gaussian_extent"]
        self.mapping_itr_num, self.gaussian_update_every = t["mapping_itr_num"], t["gaussian_update_every"]
        self.gaussian_update_offset, self.gaussian_th = t["gaussian_update_offset"], t["gaussian_th"]
        self.gaussian_extent = self.cameras_extent * t["gaussian_extent"]
        self.gaussian_reset, self.size_threshold = t["gaussian_reset"], t["size_threshold"]
        self.window_size = t["window_size"]
        self.single_thread = t.get("single_thread", True)
        self.kf_map_iters = t.get("kf_map_iters", 70)                     # run(): iter_per_kf = 70 (:934); map(iters=200) for dynamic (:1000)
        self.dynamic_map_iters = t.get("dynamic_map_iters", 200)
        self.static_map_iters = t.get("static_map_iters", 20)             # map_static(iters=20), :997
        self.network_init_iters = t.get("network_init_iters", 100)        # initialize_network: range(100), :169

    # ---- adding Gaussians (:107-128) -----------------------------------------------------------------------------------
    def add_next_kf(self, frame_idx, viewpoint, init=False, scale=2.0, depth_map=None):
        self.gaussians.extend_from_pcd_seq(viewpoint, kf_id=frame_idx, init=init, scale=scale, depthmap=depth_map)
        if self.dynamic_model and frame_idx == self.dystart:
            self.gaussians.extend_from_pcd_seq(viewpoint, kf_id=frame_idx, init=True, scale=scale, depthmap=depth_map, add_dygs=True)

    def reset(self):
        """:143-157."""
        self.iteration_count, self.occ_aware_visibility, self.viewpoints, self.current_window = 0, {}, {}, []
        self._flow_targets = {}
        self.initialized = not self.monocular
        if self.gaussians.get_xyz.shape[0]:
            self.gaussians.prune_points(self.gaussians.unique_kfIDs >= 0)

    # ---- dynamic deltas of one view -----------------------------------------------------------------------------------
    def _deltas(self, viewpoint, train=True):
        """d_values of :361-373 for the dynamic subset, or (0, 0, None)-style "no deltas" when the network is not initialised."""
        g = self.gaussians
        if not (self.dynamic_model and g.deform_init and g.dyn_rows().shape[0] > 0):
            return None, None, None
        # inside a mapping iteration (ControlNodes.begin_iteration .. end_iteration) neither the nodes nor the Gaussians move: the deltas of
        # a time are blended once and shared by the view's render and the flow terms of the views whose partner it is
        nodes = g.deform.deform
        cache = getattr(self, "_delta_cache", None) if (train and nodes._batch is not None) else None
        key = round(float(viewpoint.time), 7)
        if cache is not None and key in cache:
            return cache[key]
        time_input = nodes.expand_time(viewpoint.fid)
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            d = g.deform.step(g.get_dygs_xyz.detach(), time_input, iteration=0, feature=None, motion_mask=g.motion_mask,
                              camera_center=viewpoint.camera_center, time_interval=g.time_interval, t_key=viewpoint.time)
        out = (d["d_xyz"], d["d_scaling"], d["d_rotation"])
        if cache is not None:
            cache[key] = out
        return out

    def _render(self, viewpoint, deltas):
        dx, ds, dr = deltas
        return render(viewpoint, self.gaussians, self.pipeline_params, self.background, dynamic=False, dx=dx, ds=ds, dr=dr)

    def _render_many(self, viewpoints, deltas_list):
        """The iteration's views of this rank in one go (gaussian_renderer.render_views: one launch per pipeline stage for all of them)."""
        if len(viewpoints) == 0:
            return []
        return render_views(viewpoints, self.gaussians, self.pipeline_params, self.background,
                            deltas=[None if d[0] is None else d for d in deltas_list])

    def _view_stats(self, pkg):
        self.gaussians.add_view_stats(pkg["viewspace_points"], pkg["radii"])

    # ---- map initialisation (:237-296) -----------------------------------------------------------------------------------
    def initialize_map(self, cur_frame_idx, viewpoint):
        """:237-296. Runs of plain iterations -- between the densifications (every init_gaussian_update iterations) and the opacity resets --
        are replayed as hipGraphs (slam/mapping_graph.InitGraph: the same arithmetic, one replay per iteration)."""
        pkg = None
        rm_dynamic = not (self.dystart == cur_frame_idx)
        mapping_iteration = 0
        while mapping_iteration < self.init_itr_num:
            run = self._plain_init_run(mapping_iteration)
            if run >= self.graph_min_run and self._graphs_enabled() and self._graph_ok([viewpoint]):
                pkg = None         # (lets the last eager iteration's autograd graph go: its AccumulateGrad nodes belong to the default stream)
                done_pkg = self._initialize_map_graph_run(viewpoint, rm_dynamic, run)
                if done_pkg is not None:
                    pkg = done_pkg
                    mapping_iteration += run
                    continue
            self.iteration_count += 1
            pkg = self._render(viewpoint, (None, None, None))
            # (no name for the loss: a local that outlives the iteration keeps its autograd graph -- and the default-stream AccumulateGrad nodes of
            # the camera parameters -- alive into the next graph run's capture)
            slam_losses.get_loss_mapping(self.config, pkg["render"], pkg["depth"], viewpoint, pkg["opacity"], initialization=True,
                                         rm_dynamic=rm_dynamic, compute_value=self.loss_values).backward()
            with torch.no_grad():
                self._view_stats(pkg)
                if mapping_iteration % self.init_gaussian_update == 0:
                    self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.init_gaussian_th, self.init_gaussian_extent, None)
                if self.iteration_count == self.init_gaussian_reset or self.iteration_count == self.opt_params.densify_from_iter:
                    self.gaussians.reset_opacity()
                self.gaussians.optimizer.step()
                self.gaussians.optimizer.zero_grad(set_to_none=True)
            mapping_iteration += 1
        self.occ_aware_visibility[cur_frame_idx] = (self._final_touched(viewpoint, pkg) > 0).long()
        return pkg

    def _plain_init_run(self, mapping_iteration):
        """How many iterations of initialize_map from `mapping_iteration` on neither densify nor reset opacities."""
        n = 0
        while mapping_iteration + n < self.init_itr_num:
            count = self.iteration_count + n + 1
            if (mapping_iteration + n) % self.init_gaussian_update == 0 or count == self.init_gaussian_reset or count == self.opt_params.densify_from_iter:
                break
            n += 1
        return n

    def _initialize_map_graph_run(self, viewpoint, rm_dynamic, run):
        """`run` plain iterations of initialize_map as warm-up + capture + replays; returns the last render package, or None when nothing was
        executed (the caller goes on eagerly)."""
        from .mapping_graph import InitGraph
        from diff_gaussian_rasterization import _C
        g = self.gaussians
        stats = self.__dict__.setdefault("init_graph_stats", {"runs": 0, "replays": 0, "direct": 0, "redone": 0, "failed": 0})
        if getattr(self, "_init_graph_broken", False) or g.optimizer.scheduled_segments() is None:
            return None
        try:
            ig = InitGraph(self, viewpoint, rm_dynamic, run)
        except RuntimeError as e:
            self._graph_note(stats, e)
            return None
        warm = min(self.graph_warmup, run)
        ig.warm_up(warm)
        done = warm
        if run > warm:
            overflow0 = _C.forward_status()[0]
            ig.snapshot()
            before = _C.debug_view_slots(1)[0]
            try:
                ig.capture()
                ig.replay(run - warm)
                torch.cuda.current_stream(ig.device).synchronize()
                ok = _C.forward_status()[0] == overflow0
                if not ok:      # what outgrew what (kept in the statistics: the capture margins of slam/mapping_graph.py are sized from these)
                    after = _C.debug_view_slots(1)[0]
                    stats.setdefault("overflow_causes", []).append({"iteration": int(self.iteration_count), "run": int(run), "gaussians": int(g.get_xyz.shape[0]),
                                                                    "estimate_R": before["estimate_R_alloc"], "estimate_longest_tile": before["estimate_longest_tile"],
                                                                    "overflowing_R_alloc": after["R_alloc"], "overflowing_longest_tile": after["longest_tile"]})
            except Exception as e:        # a failed capture leaves the warm-up valid: the rest of the run goes on eagerly, no more captures
                self._init_graph_broken = True
                torch.cuda.synchronize(ig.device)
                self._graph_note(stats, e)
                ok = False
            if ok:
                done = run
                stats["replays"] += run - warm
                stats["runs"] += 1
            else:                          # a replayed frame outgrew its binning buffer (or the capture failed): undo, let the eager loop repeat
                ig.restore()
                g.optimizer.zero_grad(set_to_none=True)
                stats["redone"] += run - warm
        stats["direct"] += warm
        self.iteration_count += done
        g.optimizer.advance_steps(ig.todo, done)
        pkg = ig.pkg if done == run else None
        if done < run:                     # finish the run eagerly, iteration by iteration (plain ones: no densification inside a run)
            for _ in range(run - done):
                self.iteration_count += 1
                pkg = self._render(viewpoint, (None, None, None))
                slam_losses.get_loss_mapping(self.config, pkg["render"], pkg["depth"], viewpoint, pkg["opacity"], initialization=True,
                                             rm_dynamic=rm_dynamic, compute_value=self.loss_values).backward()
                with torch.no_grad():
                    self._view_stats(pkg)
                    g.optimizer.step()
                    g.optimizer.zero_grad(set_to_none=True)
        return pkg

    def _final_touched(self, viewpoint, pkg):
        """n_touched of the LAST render, re-rendered if the model was rebuilt after it (densification changes the row count)."""
        if pkg is not None and pkg["n_touched"].shape[0] == self.gaussians.get_xyz.shape[0]:
            return pkg["n_touched"]
        with torch.no_grad():
            return self._render(viewpoint, (None, None, None))["n_touched"]

    def initialize_network(self, cur_frame_idx, viewpoint, update_gaussians=False):
        """:160-234: create the control nodes from the keyframe's dynamic pixels and fit the node network on this view."""
        g = self.gaussians
        if g.deform is None:
            return
        if cur_frame_idx == self.dystart:
            if not g.dyn_rows().shape[0] > 0:
                return
            g.deform.extend_node_from_point(init_pcl=g.get_dygs_xyz.detach())
            g.deform_init = True
        pkg = None
        from . import dynamic_graph
        if dynamic_graph.NetworkInit.eligible(self, viewpoint, update_gaussians):
            # the normal case: the loop in the indexed layout of slam/dynamic_graph.py, its iterations after the first as hipGraph replays
            pkg = dynamic_graph.NetworkInit(self, viewpoint).run(self.network_init_iters)
            self._finish_network_init(cur_frame_idx, viewpoint, pkg)
            return
        for mapping_iteration in range(self.network_init_iters):
            deltas = self._deltas(viewpoint)
            pkg = self._render(viewpoint, deltas)
            loss_init = slam_losses.get_loss_mapping(self.config, pkg["render"], pkg["depth"], viewpoint, pkg["opacity"], initialization=True, compute_value=self.loss_values)
            loss_init.backward()
            # (every rank fits the network on this one view, redundantly: the loop is bit-reproducible since round 4 -- ordered scatter sums in
            # the node blend's backward and in the regularisers' gathers -- so the replicas stay identical without exchanging anything)
            with torch.no_grad():
                self._view_stats(pkg)
                if mapping_iteration % self.init_gaussian_update == 0:           # :209-215 (iteration 0 with the shipped schedule)
                    g.densify_and_prune(self.opt_params.densify_grad_threshold, self.init_gaussian_th, self.init_gaussian_extent, None)
                g.deform.optimizer.step()
                g.deform.optimizer.zero_grad(set_to_none=True)
                if update_gaussians:
                    g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
        self._finish_network_init(cur_frame_idx, viewpoint, pkg)

    def _finish_network_init(self, cur_frame_idx, viewpoint, pkg):
        if pkg is None or pkg["n_touched"].shape[0] != self.gaussians.get_xyz.shape[0]:      # the model was rebuilt after the last render
            with torch.no_grad():
                pkg = self._render(viewpoint, self._deltas(viewpoint, train=False))
        self.occ_aware_visibility[cur_frame_idx] = (pkg["n_touched"] > 0).long()

    # ---- window optimisation --------------------------------------------------------------------------------------------
    def _pose_updates(self, viewpoint_stack, current_window, positions=None):
        """keyframe_optimizers.step() + update_pose of :748-755 / :1213-1222: ONE camera-step launch for the window keyframes this rank owns
        (Camera.pose_steps; objects without it -- test doubles -- are stepped one by one). positions (map(): the views are key_opt, not the
        window): per view its index in the window or None -- a view outside the window has no optimizer in the reference (:955-992) and
        only drops its camera gradients; the pose is optimised for the views whose WINDOW position is below frames_to_optimize."""
        lr = self.config["Training"]["lr"]
        batch = []
        for cam_idx in range(min(len(current_window), len(viewpoint_stack))):
            viewpoint = viewpoint_stack[cam_idx]
            if not self.shard.owns(cam_idx):          # another rank rendered this view: it holds the gradient and takes the step
                continue
            position = cam_idx if positions is None else positions[cam_idx]
            if viewpoint.uid == 0 or position is None:
                for p in (viewpoint.cam_rot_delta, viewpoint.cam_trans_delta, viewpoint.exposure_a, viewpoint.exposure_b):
                    if p is not None:
                        p.grad = None
                continue
            req = (viewpoint, lr["cam_rot_delta"] * self.pose_lr_scale, lr["cam_trans_delta"] * self.pose_lr_scale, 0.01,
                   position < self.frames_to_optimize, True)
            if hasattr(type(viewpoint), "pose_steps"):
                batch.append(req)
            else:
                viewpoint.pose_step(req[1], req[2], req[3], optimize_pose=req[4], optimize_exposure=True)
        if batch:
            type(batch[0][0]).pose_steps(batch)
        self.shard.sync_cameras(viewpoint_stack[:min(len(current_window), len(viewpoint_stack))])

    def _publish_visibility(self, current_window, rows, n_views=None):
        """occ_aware_visibility of the window keyframes (:661-665) from the rows their owners hold. n_views (map(): fewer optimised views
        than window entries, where the reference's loop would run off its list): the window entries beyond it keep their previous row
        when it still fits the model, else an empty one."""
        n_views = len(current_window) if n_views is None else min(n_views, len(current_window))
        rows = {k: (n_touched > 0).long() for k, n_touched in rows.items() if k < n_views}
        like = torch.zeros(self.gaussians.get_xyz.shape[0], dtype=torch.int64, device=self.gaussians.get_xyz.device)
        if rows:
            like = torch.zeros_like(next(iter(rows.values())))
        full = self.shard.gather_mask_rows(rows, n_views, int(like.shape[0]), like.device)       # 1 bit per Gaussian and keyframe on the wire
        previous = self.occ_aware_visibility
        self.occ_aware_visibility = {current_window[idx]: full[idx] for idx in range(n_views)}
        for kf_idx in current_window[n_views:]:
            old = previous.get(kf_idx)
            self.occ_aware_visibility[kf_idx] = old if (old is not None and old.shape == like.shape) else torch.zeros_like(like)

    def _reset_opacity_of_unseen(self, pkgs):
        """reset_opacity_nonvisible (:722-728) with the visibility filters of ALL views of the iteration, whoever rendered them."""
        g = self.gaussians
        seen = torch.zeros(g.get_xyz.shape[0], dtype=torch.bool, device=g.get_xyz.device)
        for p in pkgs:
            seen |= p["visibility_filter"]
        g.reset_opacity_nonvisible([self.shard.union(seen)])

    def _window_full_bookkeeping(self, current_window):
        """What remains of the reference's covisibility pruning (:663-696 / :1140-1170) for RGB-D input: once the window is full, the
        per-Gaussian observation counts are refreshed (one stacked reduction over the window's visibility sets) and the map counts as
        initialised. The pruning itself applies to monocular input only, which this back-end does not take (slam/frontend.py refuses it)."""
        if len(current_window) != self.config["Training"]["window_size"]:
            return
        g = self.gaussians
        if self.occ_aware_visibility:
            g.n_obs.copy_(torch.stack([v != 0 for v in self.occ_aware_visibility.values()]).sum(dim=0).to(g.n_obs.dtype))
        self.initialized = True

    def _regulariser_weights(self, n_window, n_random):
        """1e-3 per window view, 1e-4 per random keyframe (:520-524,:649-652) as a device vector, built once per (window size, extras):
        a torch.tensor(..., device=) per iteration is a blocking pageable copy (~0.2 ms behind a busy queue)."""
        cache = self.__dict__.setdefault("_reg_weight_cache", {})
        if (n_window, n_random) not in cache:
            cache[(n_window, n_random)] = torch.tensor([1e-3] * n_window + [1e-4] * n_random, dtype=torch.float32, device=self.device)
        return cache[(n_window, n_random)]

    def _isotropic_loss(self):
        g = self.gaussians
        raw = g._scaling
        if (raw.is_cuda and raw.dtype == torch.float32 and raw.dim() == 2 and raw.shape[1] == 3 and g.scaling_activation is torch.exp
                and os.environ.get("GSR_FUSED_ISOTROPIC", "1") != "0"):
            return _IsotropicLoss.apply(raw)                                                # two launches forward, one back
        scaling = g.get_scaling
        return 10 * torch.abs(scaling - scaling.mean(dim=1).view(-1, 1)).mean()            # :653-655

    def map_static(self, current_window, prune=False, iters=1):
        """:1013-1224. Runs of plain iterations -- no densification, no opacity reset, one process -- go through a captured hipGraph
        (slam/mapping_graph.py: same arithmetic, one replay per iteration instead of ~55 launches from Python); ``Training.mapping_graph =
        False`` keeps every iteration eager."""
        if len(current_window) == 0:
            return
        viewpoint_stack = [self.viewpoints[kf_idx] for kf_idx in current_window]
        window_set = set(current_window)
        random_viewpoint_stack = [v for idx, v in self.viewpoints.items() if idx not in window_set]
        gaussian_split = False
        it = 0
        while it < iters:
            run = self._plain_run_length(it, iters, prune)
            if run >= self.graph_min_run and self._graph_ok(viewpoint_stack):
                done = self._map_static_graph_run(current_window, viewpoint_stack, random_viewpoint_stack, run, last=(it + run == iters))
                if done:
                    it += done
                    gaussian_split = False
                    continue
            gaussian_split = self._map_static_iteration(current_window, viewpoint_stack, random_viewpoint_stack, prune, last=(it == iters - 1))
            if prune:
                return False
            it += 1
        return gaussian_split

    def _draw_extras(self, n_candidates):
        """The two random keyframes of an iteration (:1031-1037), as indices into the non-window keyframes; the same draw on every rank."""
        return [int(c) for c in torch.randperm(n_candidates)[:2]]

    def _map_static_iteration(self, current_window, viewpoint_stack, random_viewpoint_stack, prune, last, extras_idx=None):
        """One iteration of :1013-1224, launched piece by piece."""
        shard = self.shard
        self.iteration_count += 1
        self.last_sent += 1
        loss_mapping = 0
        pkgs, touched_rows = [], {}
        if extras_idx is None:
            extras_idx = self._draw_extras(len(random_viewpoint_stack))
        extras = [random_viewpoint_stack[c] for c in extras_idx]
        mine = [(k, viewpoint) for k, viewpoint in enumerate(viewpoint_stack + extras) if shard.owns(k)]
        rendered = self._render_many([v for _, v in mine], [(None, None, None)] * len(mine))
        for (k, viewpoint), pkg in zip(mine, rendered):
            loss = slam_losses.get_loss_mapping(self.config, pkg["render"], pkg["depth"], viewpoint, pkg["opacity"], rm_dynamic=True,
                                                compute_value=self.loss_values)
            pkgs.append(pkg)
            if k < len(viewpoint_stack):
                touched_rows[k] = pkg["n_touched"]            # (turned into a 0 / 1 row when it is published)
            loss_mapping = loss_mapping + loss        # ONE backward for all views: the multi-view backward pass takes them together
        if shard.rank == 0:
            loss_mapping = loss_mapping + self._isotropic_loss()
        if torch.is_tensor(loss_mapping) and loss_mapping.requires_grad:
            loss_mapping.backward()
        shard.reduce_gradients(self.gaussians.optimizer)
        gaussian_split = False
        with torch.no_grad():
            if prune or last:                      # (every iteration overwrites the previous one's rows: only the last ones are ever read)
                self._publish_visibility(current_window, touched_rows)
            if prune:
                self._window_full_bookkeeping(current_window)
                self.gaussians.optimizer.zero_grad(set_to_none=True)
                self._clear_camera_grads(viewpoint_stack)
                return False
            for pkg in pkgs:
                self._view_stats(pkg)
            update_gaussian = self.iteration_count % self.gaussian_update_every == self.gaussian_update_offset
            if update_gaussian:
                shard.reduce_statistics(self.gaussians)
                self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.gaussian_th, self.gaussian_extent, self.size_threshold)
                gaussian_split = True
            if (self.iteration_count % self.gaussian_reset) == 0 and not update_gaussian:
                self._reset_opacity_of_unseen(pkgs)
                gaussian_split = True
            self.gaussians.optimizer.step()                 # rebuilt parameters have no gradient yet and are skipped, like in the reference
            self.gaussians.optimizer.zero_grad(set_to_none=True)
            self.gaussians.update_learning_rate(self.iteration_count)
            self._pose_updates(viewpoint_stack, current_window)
            self._clear_camera_grads(extras)
        return gaussian_split

    # ---- runs of plain iterations as hipGraph replays (slam/mapping_graph.py) ---------------------------------------------------------------
    graph_min_run = 6           # a capture costs about two eager iterations of host time: shorter runs stay eager
    graph_warmup = 2            # iterations of a run executed directly before the capture (the first may still go view by view, gsr_forward_views)
    dynamic_graph_warmup = 1    # the same for a run of the dynamic call (slam/dynamic_graph.py): its views have capacity estimates from the calls before

    def _plain_run_length(self, it, iters, prune):
        """How many iterations from `it` on neither densify nor reset opacities (those replace the model's tensors and run eagerly)."""
        if prune or not self._graphs_enabled():
            return 0
        n = 0
        while it + n < iters:
            count = self.iteration_count + n + 1
            if count % self.gaussian_update_every == self.gaussian_update_offset or count % self.gaussian_reset == 0:
                break
            n += 1
        return n

    def _graphs_enabled(self):
        cfg = self.config["Training"].get("mapping_graph", True)
        return bool(cfg) and not self.shard.active and not self.loss_values and str(self.device).startswith("cuda") and not getattr(self, "_graph_broken", False)

    def _graph_ok(self, viewpoint_stack):
        from .camera import Camera
        g = self.gaussians
        return (all(isinstance(v, Camera) and v.depth is not None for v in viewpoint_stack) and g.get_xyz.shape[0] > 0
                and getattr(g.optimizer, "_fused_acc", False) and not self.config["Training"].get("monocular", False))

    def graph_streams(self, dev):
        """(warm-up stream, capture stream) of the mapping graphs, created once per device."""
        cache = self.__dict__.setdefault("_graph_streams", {})
        if dev not in cache:
            cache[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return cache[dev]

    def unit_gradient(self, dev):
        """A scalar 1 on `dev`: the gradient the graph iterations hand every loss term (torch.autograd.backward with several roots)."""
        cache = self.__dict__.setdefault("_unit_gradients", {})
        if dev not in cache:
            cache[dev] = torch.ones((), dtype=torch.float32, device=dev)
        return cache[dev]

    def graph_pool(self, dev):
        cache = self.__dict__.setdefault("_graph_pools", {})
        if dev not in cache:
            cache[dev] = torch.cuda.graph_pool_handle()
        return cache[dev]

    @property
    def keyframe_operands(self):
        if getattr(self, "_kf_operands", None) is None:
            from .mapping_graph import KeyframeOperands
            self._kf_operands = KeyframeOperands()
        return self._kf_operands

    def _map_static_graph_run(self, current_window, viewpoint_stack, random_viewpoint_stack, run, last):
        """`run` plain iterations: draws up front, `graph_warmup` iterations executed directly, one capture, run - warm-up replays. Returns
        the number of iterations done (0: nothing was executed, the caller goes on eagerly)."""
        from .mapping_graph import MappingGraph
        from diff_gaussian_rasterization import _C
        g = self.gaussians
        stats = self.__dict__.setdefault("graph_stats", {"runs": 0, "replays": 0, "direct": 0, "redone": 0, "failed": 0})
        if g.optimizer.scheduled_segments() is None:         # (no gradient buffers / moments yet: the eager iteration creates them)
            return 0
        rng_state = torch.get_rng_state()
        draws = [self._draw_extras(len(random_viewpoint_stack)) for _ in range(run)]
        count0 = self.iteration_count
        try:
            mg = MappingGraph(self, current_window, viewpoint_stack, random_viewpoint_stack, draws, count0)
        except RuntimeError as e:
            torch.set_rng_state(rng_state)
            self._graph_note(stats, e)
            return 0
        warm = min(self.graph_warmup, run)
        mg.warm_up(warm)
        overflow0 = _C.forward_status_views()     # (the directly executed iterations waited for their headers and redid what overflowed themselves)
        if run > warm:
            mg.snapshot()
            try:
                t_cap = time.perf_counter()
                mg.capture()
                stats["capture_ms"] = stats.get("capture_ms", 0.0) + (time.perf_counter() - t_cap) * 1e3
                stats["last_capture_parts_ms"] = [round(v, 3) for v in getattr(mg, "capture_parts_ms", ())]     # begin, host pass, end + instantiate, hipMallocs
            except Exception as e:        # a capture that fails leaves the iterations done so far valid: finish the run eagerly, stop capturing
                self._graph_broken = True
                torch.cuda.synchronize(mg.device)
                mg.restore()
                g.optimizer.zero_grad(set_to_none=True)
                self._clear_camera_grads(list(viewpoint_stack) + mg.slots)
                self._graph_note(stats, e)
                self._finish_run(mg, draws, mg.executed)
                self._run_eagerly(current_window, viewpoint_stack, random_viewpoint_stack, draws[mg.executed:], last)
                return run
            mg.replay(run - warm)
            torch.cuda.current_stream(mg.device).synchronize()
            if _C.forward_status_views() != overflow0:          # a replayed view outgrew its binning buffer: undo the replays, redo them eagerly
                mg.restore()
                stats["redone"] += run - warm
                self._finish_run(mg, draws, warm)
                self._run_eagerly(current_window, viewpoint_stack, random_viewpoint_stack, draws[warm:], last)
                mg.release()
                return run
        stats["runs"] += 1
        stats["direct"] += warm
        stats["replays"] += run - warm
        self._finish_run(mg, draws, run)
        if last:
            with torch.no_grad():
                self._publish_visibility(current_window, {k: mg.pkgs[k]["n_touched"] for k in range(len(viewpoint_stack))})
        mg.release()
        return run

    def _finish_run(self, mg, draws, n):
        """Host-side state after `n` graph iterations: what n eager iterations would have left (counters, Adam's step counts, learning rate)."""
        g = self.gaussians
        self.iteration_count += n
        self.last_sent += n
        g.optimizer.advance_steps(mg.todo, n)
        if n:
            g.update_learning_rate(self.iteration_count)

    def _run_eagerly(self, current_window, viewpoint_stack, random_viewpoint_stack, draws, last):
        for j, extras_idx in enumerate(draws):
            self._map_static_iteration(current_window, viewpoint_stack, random_viewpoint_stack, False, last and j == len(draws) - 1, extras_idx=extras_idx)

    def _graph_note(self, stats, error):
        stats["failed"] += 1
        stats["last_error"] = f"{type(error).__name__}: {error}"
        if self.config["Training"].get("mapping_graph") == "strict":
            raise error

    def _clear_camera_grads(self, cams):
        for v in cams:
            for p in (v.cam_rot_delta, v.cam_trans_delta, v.exposure_a, v.exposure_b):
                if p is not None:
                    p.grad = None

    def map(self, current_window, prune=False, iters=1, dynamic_network=False):
        """:306-774 with the control-node warp on the dynamic subset. The optical-flow term of the reference (:479-509) needs RAFT;
        when the dataset can supply a flow (``dataset.gt_flow``) the same term is formed with render_flow, otherwise it is skipped.
        The optimised views are the reference's ``key_opt`` (:310-318): the three newest window keyframes plus up to five older keyframes
        chosen by their overlap with the newest one (slam/keyframes.keyframe_selection_overlap), of which the loop walks the first
        len(current_window) (:357-358); the two random views per iteration come from the complement of key_opt. Two consequences the
        reference has and this keeps: the covisibility row published under ``current_window[idx]`` is the one of view ``key_opt[idx]``
        (:661-665), and only views that sit in the window have camera parameters to step (the keyframe optimizers are built from the
        window, :955-992) -- a selected keyframe outside the window is rendered for the map's sake alone."""
        if len(current_window) == 0:
            return
        g = self.gaussians
        key_opt = list(current_window[:3])
        if len(current_window) > 3:
            from .keyframes import keyframe_selection_overlap
            ds = self.dataset
            key_opt += keyframe_selection_overlap(self.viewpoints[current_window[0]], self.viewpoints, self.viewpoints[current_window[2]].uid,
                                                  (ds.fx, ds.fy, ds.cx, ds.cy, ds.width, ds.height))
        self.last_key_opt = list(key_opt)
        viewpoint_stack = [self.viewpoints[kf_idx] for kf_idx in key_opt][:len(current_window)]
        window_position = {kf_idx: p for p, kf_idx in enumerate(current_window)}
        key_set = set(key_opt)
        random_viewpoint_stack = [v for idx, v in self.viewpoints.items() if idx not in key_set]
        use_net = dynamic_network and g.deform_init
        gaussian_split = False
        shard = self.shard
        if use_net and not prune:
            # the normal case -- one process, real keyframes -- runs in the fixed layout of slam/dynamic_graph.py: the same terms as the loop
            # below with the network's batch laid out by position, and its runs of plain iterations replayed as hipGraphs
            from . import dynamic_graph
            if dynamic_graph.eligible(self, viewpoint_stack, random_viewpoint_stack):
                call = dynamic_graph.DynamicMapping(self, current_window, viewpoint_stack, [window_position.get(v.uid) for v in viewpoint_stack],
                                                    random_viewpoint_stack, iters, self.network_warmup(iters))
                return call.execute()
        # The reference runs this loop for 200 iterations and lets the Gaussians (their optimizer step, densification, the iteration
        # counter) take part only after the first 100 (:337-338,:765-770): the node network warms up alone. A shorter schedule keeps that
        # proportion -- with the literal 100 a 60- or 80-iteration schedule would never step the Gaussians of a new keyframe at all
        # (round 2's demo and test schedules did exactly that: 24 dB on the dynamic sequence against 41 dB on the static one).
        warm = self.network_warmup(iters)
        net_params = [p for grp in g.deform.optimizer.param_groups for p in grp["params"]] if use_net else []
        if use_net:
            shard.attach_network(net_params)           # (sharded runs: the network's gradients live in one flat bucket, reduced in place)
        for i in range(iters):
            if i > warm:
                self.iteration_count += 1                                   # :337-338
            self.last_sent += 1
            loss_network = 0
            loss_mapping = 0
            dynamic = i < iters / 2                                          # :350-355
            t = self.config["Training"]
            flow_weight = t["flow_loss"] if dynamic else t.get("flow_loss_fine", t["flow_loss"])
            pkgs, touched_rows = [], {}
            views = list(viewpoint_stack)
            extra = [random_viewpoint_stack[c] for c in torch.randperm(len(random_viewpoint_stack))[:2]]      # the same draw on every rank
            with_flow = use_net and flow_weight > 0 and hasattr(self.dataset, "gt_flow")
            if use_net:       # every time sample this rank asks the node network for in this iteration, as one batch (deform_model.begin_iteration)
                nodes, times, sample_times, plans = g.deform.deform, [], [], []
                arap_delta = float(t.get("delta", 5)) * g.time_interval              # :325,:518
                for k, viewpoint in enumerate(views + extra):
                    # the regularisers' random time samples (:517-519 window views: ARAP with 4 samples over `delta` intervals; :646-648
                    # random views: ARAP with 2 samples over 5 intervals; elastic: 8 samples over 5 intervals for both). Drawn on every
                    # rank (the ranks' random streams stay in step); the regularisers themselves are rank 0's.
                    window = k < len(views)
                    plan = draw_loss_times(viewpoint.time, arap_delta if window else 5 * g.time_interval, 4 if window else 2, 5 * g.time_interval)
                    plans.append(plan)
                    if shard.owns(k):
                        times.append(viewpoint.time)
                    if shard.rank == 0:
                        sample_times += plan["arap"] + plan["elastic"]
                    closest = self.find_closest_keyframe(viewpoint.uid) if (with_flow and shard.owns(k)) else None
                    if closest is not None:
                        times.append(self.viewpoints[closest].time)
                nodes.begin_iteration(times, positions_only=sample_times, blend=(g.get_dygs_xyz.detach(), g.motion_mask))
                self._delta_cache = {}
                # the two regularisers for all views at once (per view: 1e-3 in the window, 1e-4 for the random keyframes)
                nv = len(views)
                if nodes.node_num >= 3 and shard.rank == 0:
                    wts = self._regulariser_weights(nv, len(extra))
                    reg = (nodes.elastic_loss_batch([p_["elastic"] for p_ in plans]) * wts).sum()
                    if nv:
                        reg = reg + (wts[:nv] * nodes.arap_loss_batch([p_["arap"] for p_ in plans[:nv]])).sum()
                    if len(extra):
                        reg = reg + (wts[nv:] * nodes.arap_loss_batch([p_["arap"] for p_ in plans[nv:]])).sum()
                    loss_network = loss_network + reg
            mine = [(k, viewpoint) for k, viewpoint in enumerate(views + extra) if shard.owns(k)]
            mine_deltas = [self._deltas(viewpoint) if use_net else (None, None, None) for _, viewpoint in mine]
            rendered = self._render_many([v for _, v in mine], mine_deltas)
            for (k, viewpoint), deltas, pkg in zip(mine, mine_deltas, rendered):
                loss_mapping = loss_mapping + slam_losses.get_loss_mapping(self.config, pkg["render"], pkg["depth"], viewpoint, pkg["opacity"],
                                                                         rm_dynamic=not dynamic_network, dynamic=dynamic if use_net else False, compute_value=self.loss_values)
                pkgs.append(pkg)
                if k < len(views):
                    touched_rows[k] = pkg["n_touched"]            # (turned into a 0 / 1 row when it is published)
            if with_flow:
                loss_network = loss_network + self._flow_losses([(viewpoint, deltas) for (_, viewpoint), deltas in zip(mine, mine_deltas)], flow_weight)
            if shard.rank == 0:
                loss_mapping = loss_mapping + self._isotropic_loss()
            total = loss_mapping + loss_network if use_net else loss_mapping
            if torch.is_tensor(total) and total.requires_grad:
                total.backward()
            if use_net:
                g.deform.deform.end_iteration()
                self._delta_cache = None
            # (the Gaussians only step after the network's warm-up, :765-770: before that their gradients are dropped unreduced)
            shard.reduce_gradients(g.optimizer if i > warm else None, net_params)
            gaussian_split = False
            with torch.no_grad():
                if prune or i == iters - 1:
                    self._publish_visibility(current_window, touched_rows, n_views=len(views))
                if prune:
                    self._window_full_bookkeeping(current_window)
                    g.optimizer.zero_grad(set_to_none=True)
                    if use_net:
                        shard.zero_network_grads(g.deform.optimizer)
                    self._clear_camera_grads(views + extra)
                    return False
                for pkg in pkgs:
                    self._view_stats(pkg)
                update_gaussian = (self.iteration_count % self.gaussian_update_every == self.gaussian_update_offset) and i > warm
                if update_gaussian:
                    shard.reduce_statistics(g)
                    g.densify_and_prune(self.opt_params.densify_grad_threshold, self.gaussian_th, self.gaussian_extent, self.size_threshold)
                    gaussian_split = True
                if (self.iteration_count % self.gaussian_reset) == 0 and not update_gaussian and i > warm:
                    self._reset_opacity_of_unseen(pkgs)
                    gaussian_split = True
                self._pose_updates(viewpoint_stack, current_window, positions=[window_position.get(v.uid) for v in viewpoint_stack])
                self._clear_camera_grads(extra)
                if use_net:
                    g.deform.optimizer.step()
                    shard.zero_network_grads(g.deform.optimizer)
                if i > warm:                                                  # :765-770
                    g.optimizer.step()
                    g.update_learning_rate(self.iteration_count)
                g.optimizer.zero_grad(set_to_none=True)
        return gaussian_split

    def network_warmup(self, iters):
        """How many iterations of a map(iters) call the node network trains alone before the Gaussians take part (their optimizer step,
        densification and the iteration counter start at iteration warm + 1). The reference hard-codes ``i > 100`` for its 200-iteration call
        (:337-338,:765-770); the same literal here for any schedule of 200 iterations or more. INTENTIONAL DEVIATION for shorter schedules
        (demos, tests): half of the call, or ``Training.network_warmup_iters`` when the configuration sets it -- with the literal 100 a
        60- or 80-iteration call would never step the Gaussians of a new keyframe at all. The one-iteration calls (prune=True) are not
        affected either way: iteration 0 never exceeds any warm-up."""
        if iters >= 200:
            return 100
        return int(self.config["Training"].get("network_warmup_iters", iters // 2))

    def find_closest_keyframe(self, uid):
        """:299-304."""
        keys = [key for key in self.viewpoints if key < uid]
        return max(keys) if keys else None

    def _flow_losses(self, views_and_deltas, flow_weight):
        """The optical-flow terms of :479-509 for the views of one iteration: the dynamic subset's projected motion between a keyframe
        and the closest earlier one, rendered by render_flow in both directions, against the dataset's flow on the moving pixels. The
        reference obtains that flow from RAFT (utils/camera_utils.py:386-417); here the dataset supplies it (slam/dataset.py gt_flow).
        All flow images of the iteration are ONE multi-view call (gaussian_renderer.render_flow_views), a pair's two L1 terms one fused
        loss (slam_losses.masked_l1)."""
        from gaussian_renderer import render_flow_views
        requests, pairs = [], []
        for viewpoint, deltas in views_and_deltas:
            closest = self.find_closest_keyframe(viewpoint.uid)
            if closest is None or deltas[0] is None:
                continue
            other = self.viewpoints[closest]
            if viewpoint.motion_mask is None or other.motion_mask is None:
                continue
            dx1, ds1, dr1 = deltas
            dx2, ds2, dr2 = self._deltas(other)
            # constants of the keyframe pair: the flow on the moving pixels and their mask in the renderer's [C,H,W] layout (:486-488,
            # :503-505 mask both sides of the difference by ~motion_mask; with a 0 / 1 mask: the masked target minus the masked rendering)
            hit = self._flow_targets.get((viewpoint.uid, closest))
            if hit is None:
                while len(self._flow_targets) >= 4 * max(1, int(self.config["Training"].get("window_size", 8))):   # ~7 MB per pair at 640x480: keep
                    self._flow_targets.pop(next(iter(self._flow_targets)))                                            # the window's, drop the oldest
                def target(flow, mask):
                    m = (~mask).to(torch.float32)[None]
                    return (flow.permute(2, 0, 1) * m).contiguous(), m
                hit = self._flow_targets[(viewpoint.uid, closest)] = \
                    target(self.dataset.gt_flow(viewpoint.uid, closest)[0], viewpoint.motion_mask) + \
                    target(self.dataset.gt_flow(closest, viewpoint.uid)[0], other.motion_mask)      # this keyframe -> the earlier one, and back
            requests.append((viewpoint, other, dx1, dx2, dr1, ds1))
            requests.append((other, viewpoint, dx2, dx1, dr2, ds2))
            pairs.append(hit)
        if not requests:
            return 0.0
        rendered = render_flow_views(self.gaussians, requests)
        loss = 0.0
        for k, (t_back, m1, t_fwd, m2) in enumerate(pairs):
            loss = loss + slam_losses.masked_l1(flow_weight, [(rendered[2 * k]["render"], t_back, m1), (rendered[2 * k + 1]["render"], t_fwd, m2)], channels=2)
        return loss

    def color_refinement(self, iteration_total=1500, views_per_iter=10, dynamic_network=None):
        """:777-858: L1 + D-SSIM (+ 0.1 depth L1, + the isotropic term) on ten random keyframes per iteration. `dynamic_network` defaults to
        the model's flag, as the reference's caller passes it (:899-900). Static form (:828-833): the moving pixels are masked out, only the
        Gaussians step. Dynamic form, once the node network is initialised (:791-802,:822-827,:855-857): every view is rendered through the
        warp WITH its graph, the loss is unmasked, each view adds 1e-4 x arap_loss(t = its time, delta_t = 5 intervals, 8 samples), and the
        network's optimizer steps beside the Gaussians'. The iteration's views and ARAP samples go through the node network as ONE batch
        (deform_model.begin_iteration: the fused trunk) -- view by view, op by op, the 512-row network cost ten forward and ten backward
        passes of library GEMMs per iteration, each weight gradient a single 256 x 256 macro tile on one CU (119 us: 1.3 s of the
        reference-schedule stand-in's 200 iterations)."""
        from .deform_model import draw_arap_times
        lam = self.opt_params.lambda_dssim
        g = self.gaussians
        if dynamic_network is None:
            dynamic_network = self.dynamic_model
        unmasked = bool(dynamic_network and self.dynamic_model and g.deform_init)
        use_net = bool(unmasked and g.dyn_rows().shape[0] > 0)
        ids = list(self.viewpoints.keys())
        shard = self.shard
        net_params = [p for grp in g.deform.optimizer.param_groups for p in grp["params"]] if use_net else []
        if use_net:
            shard.attach_network(net_params)           # (sharded runs: the network's gradients live in one flat bucket, reduced in place)
        constants = {}
        l1_scale = (1.0 - lam) + 0.1
        l1_alpha = (1.0 - lam) / l1_scale
        fused_l1 = os.environ.get("GSR_REFINE_FUSED_L1", "1") != "0" and g.get_xyz.is_cuda
        for iteration in range(1, iteration_total + 1):
            loss = 0
            cams = [self.viewpoints[idx] for idx in random.sample(ids, min(views_per_iter, len(ids)))]      # the same draw on every rank
            if use_net:
                nodes = g.deform.deform
                plans = [draw_arap_times(cam.time, 5 * g.time_interval, 8) for cam in cams]            # drawn on every rank (streams in step)
                regularise = shard.rank == 0 and nodes.node_num >= 3                                    # (the regularisers are rank 0's, as in map())
                nodes.begin_iteration([cam.time for k, cam in enumerate(cams) if shard.owns(k)],
                                      positions_only=[t for plan in plans for t in plan] if regularise else [],
                                      blend=(g.get_dygs_xyz.detach(), g.motion_mask))
                self._delta_cache = {}
                if regularise:
                    loss = loss + 1e-4 * nodes.arap_loss_batch(plans).sum()                            # :827, one term per view
            for k, cam in enumerate(cams):
                if not shard.owns(k):
                    continue
                pkg = self._render(cam, self._deltas(cam))
                image = torch.exp(cam.exposure_a) * pkg["render"] + cam.exposure_b
                # the view's constants (ground truth on the device, the depth mask and -- static form -- the motion mask as float weights): once per
                # call of this function, not per iteration
                hit = constants.get(cam.uid)
                if hit is None:
                    gt_image = cam.original_image.to(image.device)
                    gt_depth = cam.depth_device()[None]
                    mm = None if unmasked else cam.motion_mask
                    dm = (gt_depth > 0.01) if mm is None else (gt_depth > 0.01) & mm[None]
                    if len(constants) >= 512:                      # (~2.5 MB of derived weights per keyframe at 640 x 480: bounded all the same)
                        constants.pop(next(iter(constants)))
                    hit = constants[cam.uid] = (gt_image, gt_depth, mm, None if mm is None else mm.to(torch.float32).reshape(1, *gt_depth.shape[-2:]),
                                                dm.to(torch.float32))
                gt_image, gt_depth, mm, w_rgb, w_dep = hit
                if fused_l1:
                    # (1 - lam) mean|exp(a) I + b - gt| (masked: x mm) + 0.1 mean(dm |D - gt_D|) as ONE fused weighted-L1 node (slam_losses:
                    # s (a X + (1 - a) Y) with s = (1 - lam) + 0.1, a = (1 - lam) / s), instead of ~25 element-wise launches each way per view
                    loss = loss + l1_scale * slam_losses.weighted_l1_loss(pkg["render"], pkg["depth"], gt_image, gt_depth, w_rgb, w_dep, cam.exposure_a,
                                                                         cam.exposure_b, l1_alpha)
                else:
                    Ll1 = torch.abs(image - gt_image).mean() if mm is None else torch.abs(image * mm - gt_image * mm).mean()
                    loss = loss + (1.0 - lam) * Ll1 + 0.1 * torch.abs(pkg["depth"] * w_dep - gt_depth * w_dep).mean()
                loss = loss + lam * (1.0 - slam_losses.ssim(image, gt_image, mask=mm))
            if shard.rank == 0:
                loss = loss + self._isotropic_loss()
            if torch.is_tensor(loss) and loss.requires_grad:
                loss.backward()
            if use_net:
                g.deform.deform.end_iteration()
                self._delta_cache = None
            shard.reduce_gradients(g.optimizer, net_params)
            with torch.no_grad():
                g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
                g.update_learning_rate(iteration)
                if use_net:                                                   # :855-857
                    g.deform.optimizer.step()
                    shard.zero_network_grads(g.deform.optimizer)
                self._clear_camera_grads(self.viewpoints.values())

    # ---- messages of run() (:879-1010) as calls ----------------------------------------------------------------------------
    def push_to_frontend(self, tag="sync_backend"):
        """:864-876."""
        self.last_sent = 0
        keyframes = [(kf_idx, self.viewpoints[kf_idx].R.clone(), self.viewpoints[kf_idx].T.clone()) for kf_idx in self.current_window]
        return [tag, self.gaussians, self.occ_aware_visibility, keyframes]

    def handle_init(self, cur_frame_idx, viewpoint, depth_map):
        """"init", :899-914."""
        self.reset()
        self.viewpoints[cur_frame_idx] = viewpoint
        self.add_next_kf(cur_frame_idx, viewpoint, depth_map=depth_map, init=True)
        self.initialize_map(cur_frame_idx, viewpoint)
        if self.dynamic_model and self.dystart == 0:
            self.initialize_network(cur_frame_idx, viewpoint)
        self.current_window = [cur_frame_idx]
        return self.push_to_frontend("init")

    def handle_keyframe(self, cur_frame_idx, viewpoint, current_window, depth_map, add_new_gaussian=True, dynamic_render=False):
        """"keyframe", :916-1003."""
        self.viewpoints[cur_frame_idx] = viewpoint
        self.current_window = current_window
        if add_new_gaussian:
            self.add_next_kf(cur_frame_idx, viewpoint, depth_map=depth_map)
        if self.dynamic_model and self.dystart == cur_frame_idx and cur_frame_idx > 0:
            self.initialize_map(cur_frame_idx, viewpoint)
            self.initialize_network(cur_frame_idx, viewpoint)
        self.frames_to_optimize = self.config["Training"]["pose_window"]
        if not self.initialized and len(self.current_window) == self.config["Training"]["window_size"]:
            self.frames_to_optimize = self.config["Training"]["window_size"] - 1
        for kf in self.current_window:                                       # a fresh Adam for the window's cameras, :992
            if kf != 0:
                self.viewpoints[kf].reset_pose_optimizer()
        if self.dystart > cur_frame_idx or not self.dynamic_model:
            self.map_static(self.current_window, iters=self.static_map_iters)
            self.map_static(self.current_window, prune=True)
        elif add_new_gaussian:
            self.map(self.current_window, iters=self.dynamic_map_iters, dynamic_network=self.dynamic_model)
            self.map(self.current_window, prune=True, dynamic_network=self.dynamic_model)
        return self.push_to_frontend("keyframe")
