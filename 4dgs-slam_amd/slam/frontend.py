"""Tracking front-end for RGB-D input, single process -- the role of the reference's ``utils/slam_frontend.py`` FrontEnd in its
``single_thread: True`` schedule (the front-end hands every keyframe to the back-end and waits for it, :664-666). The public methods
keep the reference's names and meaning (add_new_keyframe :128-187, initialize :189-207, tracking :335-470, is_keyframe :472-499,
add_to_window :501-562, run :603-833) so that code written against it keeps working; the GUI / wandb / queue plumbing does not exist
here, the back-end is called directly.

How a frame is processed:

  track      the pose is optimised against the static Gaussians by ``slam/tracking_graph.TrackingGraph`` -- render (fused prologue,
             static rows gathered inside the kernels, Gaussians DETACHED so that the backward pass runs pose-only) -> fused tracking
             loss -> backward -> ONE camera-step launch (Adam on pose + exposure, update_pose, matrices) -- either replayed as a
             captured hipGraph (``Training.tracking_graph``) or called eagerly; the host only polls the convergence latch every
             ``converge_check_every`` iterations (the reference synchronises every iteration, :440);
  decide     one device-side statistics vector per frame (visibility overlap with the newest keyframe, distance to it) decides
             whether the frame becomes a keyframe (``slam/keyframes.py``; ONE host transfer per frame);
  insert     the window update is computed on the device and read back as two indices; the back-end maps the new keyframe.
"""
import time

import torch

from gaussian_renderer import render

from . import keyframes as kf
from .camera import Camera
from .eval_utils import eval_ate, save_gaussians


@torch.no_grad()
def render_without_grad(viewpoint, gaussians, pipe, background):
    return render(viewpoint, gaussians, pipe, background, dynamic=False)


def lower_median(x):
    """torch.median's value (the lower of the two middle elements) through a sort: Tensor.median() takes ~3 ms for an image on this
    stack, the sort 0.1 ms."""
    flat = x.reshape(-1)
    if flat.numel() == 0:
        return flat.median()                      # nan + the reference's behaviour on empty input
    return torch.sort(flat)[0][(flat.numel() - 1) // 2]


def get_median_depth(depth, opacity=None, mask=None, return_std=False):
    """utils/slam_utils.py:367-378."""
    depth = depth.detach().clone()
    valid = depth > 0
    if opacity is not None:
        valid = torch.logical_and(valid, opacity.detach() > 0.95)
    if mask is not None:
        valid = torch.logical_and(valid, mask)
    valid_depth = depth[valid]
    if return_std:
        return lower_median(valid_depth), valid_depth.std(), valid
    return lower_median(valid_depth)


class FrontEnd:
    MIN_KEYFRAME_GAP = 5          # a keyframe at the latest every five frames (utils/slam_frontend.py:739)

    def __init__(self, config: dict):
        training, self.config = config["Training"], config
        self.monocular, self.dynamic_model = training.get("monocular", False), config["model_params"]["dynamic_model"]
        self.converge_check_every = int(training.get("converge_check_every", 5))
        self.use_tracking_graph = bool(training.get("tracking_graph", False))     # slam/tracking_graph.py
        self.device = torch.device("cuda", 0)
        # wired by slam/system.py
        self.background = self.pipeline_params = self.backend = self.dataset = self.gaussians = None
        # map / trajectory state (the attribute names are the reference's: the back-end and the evaluation read them)
        self.cameras, self.kf_indices, self.current_window, self.occ_aware_visibility = {}, [], [], {}
        self.initialized, self.reset = False, True
        self.median_depth, self.use_every_n_frames, self.iteration_count = 1.0, 1, 0
        self.dynamic_objects = self.dystart = 0
        # bookkeeping
        self._tracker = None
        self.graph_stats = {"captures": 0, "replayed_frames": 0, "eager_frames": 0, "overflow_redos": 0}
        self.log = []
        self.init_done_at = None

    def set_hyperparams(self):
        """:115-126."""
        r, t = self.config["Results"], self.config["Training"]
        self.save_dir, self.save_results = r.get("save_dir"), r.get("save_results", False)
        self.save_trj, self.save_trj_kf_intv = r.get("save_trj", False), r.get("save_trj_kf_intv", 5)
        self.tracking_itr_num, self.kf_interval = t["tracking_itr_num"], t["kf_interval"]
        self.window_size, self.single_thread = t["window_size"], t.get("single_thread", True)

    @property
    def thresholds(self):
        return kf.KeyframeThresholds.from_config(self.config)

    # ---- keyframe depth (:128-187, RGB-D branch) ---------------------------------------------------------------------
    def add_new_keyframe(self, frame_idx, depth=None, opacity=None, init=False):
        if self.monocular:         # (utils/slam_frontend.py:135-178 is not part of the RGB-D configurations shipped)
            raise NotImplementedError("monocular initialisation is not built: RGB-D input only")
        self.kf_indices += [frame_idx]
        viewpoint = self.cameras[frame_idx]
        gt_img = viewpoint.original_image.to(self.device)
        usable = gt_img.sum(dim=0) > self.config["Training"]["rgb_boundary_threshold"]
        if self.dynamic_model and viewpoint.motion_mask is not None:
            usable = usable & viewpoint.motion_mask                       # :185-186: seed the static map from static pixels only
        return viewpoint.depth_device() * usable                          # :180-181 (no host round trip, unlike x[mask] = 0)

    def initialize(self, frame_idx, viewpoint):
        """:189-207."""
        self.kf_indices, self.iteration_count, self.occ_aware_visibility, self.current_window = [], 0, {}, []
        self.initialized, self.reset = not self.monocular, False
        viewpoint.update_RT(viewpoint.R_gt, viewpoint.T_gt)               # first frame at the ground-truth pose
        seed_depth = self.add_new_keyframe(frame_idx, init=True)
        self.sync_backend(self.backend.handle_init(frame_idx, viewpoint, seed_depth))

    # ---- tracking (:335-470) -------------------------------------------------------------------------------------------
    def _tracker_for_current_map(self, viewpoint):
        """The TrackingGraph of the current map version (its static slot, the detached Gaussians and -- if asked for -- the captured
        graph); rebuilt whenever the back-end replaced the model's tensors."""
        from .tracking_graph import TrackingGraph
        if self._tracker is None or self._tracker.version != TrackingGraph.model_version(self.gaussians):
            self._tracker = TrackingGraph(self.gaussians, self.pipeline_params, self.background, self.config, viewpoint)
            if self.use_tracking_graph:
                self._tracker.load(viewpoint)
                self._tracker.capture()
                self.graph_stats["captures"] += 1
        return self._tracker

    def tracking(self, frame_idx, viewpoint, last_keyframe_idx=None):
        previous = self.cameras[frame_idx - self.use_every_n_frames]
        viewpoint.update_RT(previous.R, previous.T)
        tracker = self._tracker_for_current_map(viewpoint)
        tracker.load(viewpoint)
        replayed = False
        if self.use_tracking_graph:
            _, replayed = tracker.run(self.tracking_itr_num, self.converge_check_every)
            self.graph_stats["replayed_frames" if replayed else "overflow_redos"] += 1
            if not replayed:
                tracker.load(viewpoint)          # the frame outgrew the captured graph's binning buffer: start over, eagerly
        if replayed:
            pkg = tracker.pkg                    # the captured iteration's static outputs: the last replay's render
        else:                                    # eager: the same iteration, launch by launch
            self.graph_stats["eager_frames"] += 1
            for it in range(self.tracking_itr_num):
                pkg = tracker.iteration()
                if (it + 1) % self.converge_check_every == 0 and tracker.cam.converged():
                    break
        tracker.store(viewpoint)
        self.median_depth = get_median_depth(pkg["depth"], pkg["opacity"])        # :461
        return render_without_grad(viewpoint, self.gaussians, self.pipeline_params, self.background)

    # ---- keyframe management (:472-562): the decisions of the reference, computed by slam/keyframes.py -------------------------
    def is_keyframe(self, cur_frame_idx, last_keyframe_idx, cur_frame_visibility_filter, occ_aware_visibility):
        stats = kf.frame_statistics(self.cameras[cur_frame_idx], self.cameras[last_keyframe_idx], cur_frame_visibility_filter,
                                    occ_aware_visibility[last_keyframe_idx])
        return bool(kf.keyframe_decision(stats, self.median_depth, self.thresholds))

    def add_to_window(self, cur_frame_idx, cur_frame_visibility_filter, occ_aware_visibility, window):
        thr = self.thresholds
        old = list(window)
        cutoff = thr.cutoff if self.initialized else 0.4                  # :523-529
        drop = kf.window_evictions(self.cameras[cur_frame_idx], [self.cameras[k] for k in old], cur_frame_visibility_filter,
                                   kf.visibility_matrix(occ_aware_visibility, old, cur_frame_visibility_filter), cutoff, thr.window_size)
        low, crowded = (int(v) for v in drop.tolist())                    # the one host transfer of the window update
        gone = [old[j] for j in (low, crowded) if j >= 0]
        removed = gone[-1] if gone else None
        return [cur_frame_idx] + [k for k in old if k not in gone], removed

    def sync_backend(self, data):
        """:582-590 (single process: the Gaussians are shared, not cloned)."""
        _, self.gaussians, self.occ_aware_visibility, poses = data[:4]
        for frame, R, T in poses:
            self.cameras[frame].update_RT(R.clone(), T.clone())

    def cleanup(self, frame_idx):
        self.cameras[frame_idx].clean()

    # ---- main loop (:603-833, single-thread schedule) ----------------------------------------------------------------------
    def _wants_keyframe(self, cur_frame_idx, newest_kf, frames_since_kf, visibility):
        """Keyframe decision of one tracked frame; returns (decision, overlap ratio for the log or None)."""
        stats = kf.frame_statistics(self.cameras[cur_frame_idx], self.cameras[newest_kf], visibility, self.occ_aware_visibility[newest_kf])
        by_motion = kf.keyframe_decision(stats, self.median_depth, self.thresholds)
        iou, by_motion = torch.stack([stats[0], by_motion.to(stats.dtype)]).tolist()                  # the one host transfer of the decision
        by_motion = by_motion > 0.5
        due = frames_since_kf >= self.kf_interval
        reported = None
        if len(self.current_window) < self.window_size:                   # while the window fills up only the overlap counts (:716-728)
            wanted, reported = due and iou < self.thresholds.overlap, iou
        else:
            wanted = by_motion
        if self.single_thread:
            wanted = due and wanted
        forced = cur_frame_idx - newest_kf >= self.MIN_KEYFRAME_GAP or cur_frame_idx == self.dystart
        new_object = self.dataset.dynamic_objects > self.dynamic_objects and cur_frame_idx > 0
        return bool(wanted or forced or new_object), reported

    def _insert_keyframe(self, cur_frame_idx, viewpoint, visibility, render_pkg, overlap):
        self.current_window, _ = self.add_to_window(cur_frame_idx, visibility, self.occ_aware_visibility, self.current_window)
        depth_map = self.add_new_keyframe(cur_frame_idx, depth=render_pkg["depth"], opacity=render_pkg["opacity"], init=False)
        self.sync_backend(self.backend.handle_keyframe(cur_frame_idx, viewpoint, self.current_window, depth_map, True, False))
        self.log.append(("keyframe", cur_frame_idx, overlap))
        viewpoint.clean_key()
        if self.save_results and self.save_trj and len(self.kf_indices) % self.save_trj_kf_intv == 0:
            eval_ate(self.cameras, self.kf_indices, self.save_dir, cur_frame_idx, monocular=self.monocular)

    def run(self, max_frames=None):
        n_frames = len(self.dataset) if max_frames is None else min(max_frames, len(self.dataset))
        previous_kf = 0                                   # the window's newest keyframe as of the previous frame
        for cur_frame_idx in range(n_frames):
            viewpoint = Camera.init_from_dataset(self.dataset, cur_frame_idx, self.dataset.projection_matrix)
            viewpoint.compute_grad_mask(self.config)
            self.cameras[cur_frame_idx] = viewpoint
            if self.reset:                                # first frame: build the map from it
                self.initialize(cur_frame_idx, viewpoint)
                self.current_window += [cur_frame_idx]
                torch.cuda.synchronize(self.device)
                self.init_done_at = time.perf_counter()   # map initialisation (hundreds of iterations on one frame) is reported apart
                continue
            self.initialized = self.initialized or len(self.current_window) == self.window_size
            render_pkg = self.tracking(cur_frame_idx, viewpoint, previous_kf)
            frames_since_kf = cur_frame_idx - previous_kf
            previous_kf = newest_kf = self.current_window[0]
            visibility = (render_pkg["n_touched"] > 0).long()
            wanted, overlap = self._wants_keyframe(cur_frame_idx, newest_kf, frames_since_kf, visibility)
            if wanted:
                self._insert_keyframe(cur_frame_idx, viewpoint, visibility, render_pkg, overlap)
            else:
                self.cleanup(cur_frame_idx)
            self.dynamic_objects = self.dataset.dynamic_objects
        if self.save_results and self.save_dir:
            eval_ate(self.cameras, self.kf_indices, self.save_dir, 0, final=True, monocular=self.monocular)
            save_gaussians(self.gaussians, self.save_dir, "final", final=True)
