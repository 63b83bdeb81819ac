"""Tracking front-end -- the counterpart of the reference's ``utils/slam_frontend.py`` FrontEnd for RGB-D input, single process
(its ``single_thread: True`` schedule: the front-end waits for the back-end after every keyframe, :664-666). Same methods and the
same decisions: add_new_keyframe (:128-187), initialize (:189-207), tracking (:335-470), is_keyframe (:472-499), add_to_window
(:501-562), the run loop (:603-833). GUI / wandb / queue plumbing is out of scope; the back-end is called directly.

What changed underneath: every tracking iteration is render (fused prologue, static Gaussians gathered inside the kernels) -> fused
tracking loss -> backward -> ONE camera-step launch (Adam on pose + exposure, update_pose, matrices; Camera.pose_step). The only
host synchronisations per frame are the convergence poll (every ``converge_check_every`` iterations; the reference synchronises
every iteration, :440) and the median depth / visibility reads of the keyframe test."""
import numpy as np
import time

import torch

from gaussian_renderer import render
import slam_losses
from diff_gaussian_rasterization import raw as _raw

from .camera import Camera
from .eval_utils import eval_ate, save_gaussians


def getWorld2View2(R, t):
    """gaussian_splatting/utils/graphics_utils.py:38-50 with the default translate / scale."""
    Rt = torch.zeros((4, 4), device=R.device)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt


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
    def __init__(self, config):
        self.config = config
        self.background = None
        self.pipeline_params = None
        self.backend = None
        self.dataset = None
        self.initialized = False
        self.kf_indices = []
        self.monocular = config["Training"].get("monocular", False)
        self.iteration_count = 0
        self.occ_aware_visibility = {}
        self.current_window = []
        self.reset = True
        self.use_every_n_frames = 1
        self.gaussians = None
        self.cameras = dict()
        self.device = "cuda:0"
        self.dynamic_model = config["model_params"]["dynamic_model"]
        self.dynamic_objects = 0
        self.dystart = 0
        self.median_depth = 1.0
        self.converge_check_every = int(config["Training"].get("converge_check_every", 5))
        self.use_tracking_graph = bool(config["Training"].get("tracking_graph", False))     # slam/tracking_graph.py
        self._tgraph = None
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

    # ---- keyframe depth (:128-187, RGB-D branch) ---------------------------------------------------------------------
    def add_new_keyframe(self, cur_frame_idx, depth=None, opacity=None, init=False):
        if self.monocular:
            raise NotImplementedError("monocular initialisation (utils/slam_frontend.py:135-178) is not part of the RGB-D configurations shipped")
        self.kf_indices.append(cur_frame_idx)
        viewpoint = self.cameras[cur_frame_idx]
        gt_img = viewpoint.original_image.to(self.device)
        valid_rgb = (gt_img.sum(dim=0) > self.config["Training"]["rgb_boundary_threshold"])
        initial_depth = viewpoint.depth_device().clone()
        initial_depth.masked_fill_(~valid_rgb, 0)                           # :180-181 (masked_fill: no host round trip, unlike x[mask] = 0)
        if self.dynamic_model and viewpoint.motion_mask is not None:
            initial_depth.masked_fill_(~viewpoint.motion_mask, 0)           # :185-186: seed the static map from static pixels only
        return initial_depth

    def initialize(self, cur_frame_idx, viewpoint):
        """:189-207."""
        self.initialized = not self.monocular
        self.kf_indices, self.iteration_count, self.occ_aware_visibility, self.current_window = [], 0, {}, []
        viewpoint.update_RT(viewpoint.R_gt, viewpoint.T_gt)               # first frame at the ground-truth pose
        depth_map = self.add_new_keyframe(cur_frame_idx, init=True)
        self.sync_backend(self.backend.handle_init(cur_frame_idx, viewpoint, depth_map))
        self.reset = False

    # ---- tracking (:335-470) -------------------------------------------------------------------------------------------
    def _track_with_graph(self, viewpoint):
        """The tracking loop as hipGraph replays (slam/tracking_graph.py). Returns False if the frame has to be redone eagerly."""
        from .tracking_graph import TrackingGraph
        if self._tgraph is None or self._tgraph.version != TrackingGraph.model_version(self.gaussians):
            self._tgraph = TrackingGraph(self.gaussians, self.pipeline_params, self.background, self.config, viewpoint)
            self._tgraph.load(viewpoint)
            self._tgraph.capture()
            self.graph_stats["captures"] += 1
        self._tgraph.load(viewpoint)
        _, ok = self._tgraph.run(self.tracking_itr_num, self.converge_check_every)
        if ok:
            self._tgraph.store(viewpoint)
            self.graph_stats["replayed_frames"] += 1
        else:
            self.graph_stats["overflow_redos"] += 1
        return ok

    def tracking(self, cur_frame_idx, viewpoint, last_keyframe_idx):
        prev = self.cameras[cur_frame_idx - self.use_every_n_frames]
        viewpoint.update_RT(prev.R, prev.T)
        if self.use_tracking_graph and self._track_with_graph(viewpoint):
            self.median_depth = get_median_depth(self._tgraph.pkg["depth"], self._tgraph.pkg["opacity"])      # the last tracking iteration's render, :461
            with torch.no_grad():
                render_pkg = render(viewpoint, self.gaussians, self.pipeline_params, self.background, dynamic=False)
            return render_pkg
        self.graph_stats["eager_frames"] += 1
        lr = self.config["Training"]["lr"]
        viewpoint.reset_pose_optimizer()
        static = None
        if self.gaussians.dyn_rows().shape[0] > 0:
            static = self.gaussians.dygs == False  # noqa: E712  (the reference's expression, :413)
            static._gsr_gather = _raw.gather_from_mask(static)             # one nonzero() per frame instead of one per iteration
        depth = opacity = None
        for tracking_itr in range(self.tracking_itr_num):
            render_pkg = render(viewpoint, self.gaussians, self.pipeline_params, self.background, dynamic=False, mask=static)
            image, depth, opacity = render_pkg["render"], render_pkg["depth"], render_pkg["opacity"]
            loss_tracking = slam_losses.get_loss_tracking(self.config, image, depth, opacity, viewpoint, rm_dynamic=True, mask=None)
            loss_tracking.backward()
            viewpoint.pose_step(lr["cam_rot_delta"], lr["cam_trans_delta"], 0.01, latch=True)      # step + zero_grad + update_pose, :434-440
            self.gaussians.optimizer.zero_grad(set_to_none=True)
            if (tracking_itr + 1) % self.converge_check_every == 0 and viewpoint.converged():
                break
        self.median_depth = get_median_depth(depth, opacity)
        with torch.no_grad():
            render_pkg = render(viewpoint, self.gaussians, self.pipeline_params, self.background, dynamic=False)
        return render_pkg

    # ---- keyframe management (:472-562), restated decision by decision ------------------------------------------------
    def is_keyframe(self, cur_frame_idx, last_keyframe_idx, cur_frame_visibility_filter, occ_aware_visibility):
        t = self.config["Training"]
        curr_frame, last_kf = self.cameras[cur_frame_idx], self.cameras[last_keyframe_idx]
        pose_CW = getWorld2View2(curr_frame.R, curr_frame.T)
        last_kf_WC = torch.linalg.inv(getWorld2View2(last_kf.R, last_kf.T))
        dist = torch.norm((pose_CW @ last_kf_WC)[0:3, 3])
        dist_check = dist > t["kf_translation"] * self.median_depth
        dist_check2 = dist > t["kf_min_translation"] * self.median_depth
        union = torch.logical_or(cur_frame_visibility_filter, occ_aware_visibility[last_keyframe_idx]).count_nonzero()
        intersection = torch.logical_and(cur_frame_visibility_filter, occ_aware_visibility[last_keyframe_idx]).count_nonzero()
        point_ratio_2 = intersection / union
        return bool((point_ratio_2 < t["kf_overlap"] and dist_check2) or dist_check)

    def add_to_window(self, cur_frame_idx, cur_frame_visibility_filter, occ_aware_visibility, window):
        N_dont_touch = 2
        window = [cur_frame_idx] + window
        curr_frame = self.cameras[cur_frame_idx]
        to_remove = []
        removed_frame = None
        for i in range(N_dont_touch, len(window)):
            kf_idx = window[i]
            intersection = torch.logical_and(cur_frame_visibility_filter, occ_aware_visibility[kf_idx]).count_nonzero()      # Szymkiewicz-Simpson
            denom = min(cur_frame_visibility_filter.count_nonzero(), occ_aware_visibility[kf_idx].count_nonzero())
            point_ratio_2 = intersection / denom
            cut_off = self.config["Training"]["kf_cutoff"] if "kf_cutoff" in self.config["Training"] else 0.4
            if not self.initialized:
                cut_off = 0.4
            if point_ratio_2 <= cut_off:
                to_remove.append(kf_idx)
        if to_remove:
            window.remove(to_remove[-1])
            removed_frame = to_remove[-1]
        kf_0_WC = torch.linalg.inv(getWorld2View2(curr_frame.R, curr_frame.T))
        if len(window) > self.config["Training"]["window_size"]:
            inv_dist = []
            for i in range(N_dont_touch, len(window)):
                inv_dists = []
                kf_i = self.cameras[window[i]]
                kf_i_CW = getWorld2View2(kf_i.R, kf_i.T)
                for j in range(N_dont_touch, len(window)):
                    if i == j:
                        continue
                    kf_j = self.cameras[window[j]]
                    T_CiCj = kf_i_CW @ torch.linalg.inv(getWorld2View2(kf_j.R, kf_j.T))
                    inv_dists.append(1.0 / (torch.norm(T_CiCj[0:3, 3]) + 1e-6).item())
                T_CiC0 = kf_i_CW @ kf_0_WC
                k = torch.sqrt(torch.norm(T_CiC0[0:3, 3])).item()
                inv_dist.append(k * sum(inv_dists))
            idx = int(np.argmax(inv_dist))
            removed_frame = window[N_dont_touch + idx]
            window.remove(removed_frame)
        return window, removed_frame

    def sync_backend(self, data):
        """:582-590 (single process: the Gaussians are shared, not cloned)."""
        self.gaussians = data[1]
        self.occ_aware_visibility = data[2]
        for kf_id, kf_R, kf_T in data[3]:
            self.cameras[kf_id].update_RT(kf_R.clone(), kf_T.clone())

    def cleanup(self, cur_frame_idx):
        self.cameras[cur_frame_idx].clean()

    # ---- main loop (:603-833, single-thread schedule) ----------------------------------------------------------------------
    def run(self, max_frames=None):
        cur_frame_idx, last_keyframe_idx = 0, 0
        projection_matrix = self.dataset.projection_matrix
        n_frames = len(self.dataset) if max_frames is None else min(max_frames, len(self.dataset))
        while cur_frame_idx < n_frames:
            viewpoint = Camera.init_from_dataset(self.dataset, cur_frame_idx, projection_matrix)
            viewpoint.compute_grad_mask(self.config)
            self.cameras[cur_frame_idx] = viewpoint
            if self.reset:
                self.initialize(cur_frame_idx, viewpoint)
                self.current_window.append(cur_frame_idx)
                cur_frame_idx += 1
                torch.cuda.synchronize(self.device)
                self.init_done_at = time.perf_counter()       # map initialisation (hundreds of iterations on one frame) is reported apart
                continue
            self.initialized = self.initialized or (len(self.current_window) == self.window_size)
            render_pkg = self.tracking(cur_frame_idx, viewpoint, last_keyframe_idx)
            check_time = (cur_frame_idx - last_keyframe_idx) >= self.kf_interval
            last_keyframe_idx = self.current_window[0]
            curr_visibility = (render_pkg["n_touched"] > 0).long()
            create_kf = self.is_keyframe(cur_frame_idx, last_keyframe_idx, curr_visibility, self.occ_aware_visibility)
            point_ratio = None
            if len(self.current_window) < self.window_size:
                union = torch.logical_or(curr_visibility, self.occ_aware_visibility[last_keyframe_idx]).count_nonzero()
                intersection = torch.logical_and(curr_visibility, self.occ_aware_visibility[last_keyframe_idx]).count_nonzero()
                point_ratio = intersection / union
                create_kf = bool(check_time and point_ratio < self.config["Training"]["kf_overlap"])
            if self.single_thread:
                create_kf = check_time and create_kf
            create_kf = ((cur_frame_idx - last_keyframe_idx) >= 5) or create_kf or cur_frame_idx == self.dystart        # :739
            if self.dataset.dynamic_objects > self.dynamic_objects and cur_frame_idx > 0:
                create_kf = True
            if create_kf:
                self.current_window, removed = self.add_to_window(cur_frame_idx, curr_visibility, self.occ_aware_visibility, self.current_window)
                depth_map = self.add_new_keyframe(cur_frame_idx, depth=render_pkg["depth"], opacity=render_pkg["opacity"], init=False)
                self.sync_backend(self.backend.handle_keyframe(cur_frame_idx, viewpoint, self.current_window, depth_map, True, False))
                self.log.append(("keyframe", cur_frame_idx, None if point_ratio is None else float(point_ratio)))
                self.cameras[cur_frame_idx].clean_key()
                if self.save_results and self.save_trj and len(self.kf_indices) % self.save_trj_kf_intv == 0:
                    eval_ate(self.cameras, self.kf_indices, self.save_dir, cur_frame_idx, monocular=self.monocular)
            else:
                self.cleanup(cur_frame_idx)
            cur_frame_idx += 1
            self.dynamic_objects = self.dataset.dynamic_objects
        if self.save_results and self.save_dir:
            eval_ate(self.cameras, self.kf_indices, self.save_dir, 0, final=True, monocular=self.monocular)
            save_gaussians(self.gaussians, self.save_dir, "final", final=True)
