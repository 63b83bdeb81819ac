"""The SLAM system -- the counterpart of the reference's ``slam.py`` SLAM class (:39-230) in one process: builds the Gaussian model,
front-end and back-end from a configuration with the reference's YAML structure (configs/rgbd/tum/base_config.yaml), runs the
sequence, evaluates ATE / PSNR (utils/eval_utils.py). GUI, wandb, multiprocessing queues and the dataset loaders are out of scope;
any object with the interface of slam/dataset.py can be passed as the dataset."""
import copy
import time
import types

import torch

from .backend import BackEnd
from .deform_model import DeformModel
from .eval_utils import eval_ate, eval_rendering, save_gaussians
from .frontend import FrontEnd
from .gaussian_model import GaussianModel


def default_config():
    """configs/rgbd/tum/base_config.yaml, value by value."""
    return {
        "Results": {"save_results": False, "save_dir": None, "save_trj": False, "save_trj_kf_intv": 5, "use_gui": False, "eval_rendering": True,
                    "use_wandb": False},
        "Dataset": {"type": "synthetic", "sensor_type": "depth", "pcd_downsample": 128, "pcd_downsample_init": 32, "adaptive_pointsize": True,
                    "point_size": 0.01},
        "Training": {"init_itr_num": 1050, "init_gaussian_update": 100, "init_gaussian_reset": 500, "init_gaussian_th": 0.005,
                     "init_gaussian_extent": 30, "tracking_itr_num": 100, "mapping_itr_num": 50, "gaussian_update_every": 150,
                     "gaussian_update_offset": 50, "gaussian_th": 0.7, "gaussian_extent": 1.0, "gaussian_reset": 2001, "size_threshold": 20,
                     "kf_interval": 5, "window_size": 8, "pose_window": 3, "edge_threshold": 1.1, "rgb_boundary_threshold": 0.01, "alpha": 0.9,
                     "kf_translation": 0.08, "kf_min_translation": 0.05, "kf_overlap": 0.9, "kf_cutoff": 0.3, "prune_mode": "slam",
                     "single_thread": True, "spherical_harmonics": False, "flow_loss": 3, "monocular": False,
                     "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}},
        "opt_params": {"iterations": 30000, "position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_delay_mult": 0.01,
                       "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05, "scaling_lr": 0.001, "rotation_lr": 0.001,
                       "percent_dense": 0.01, "lambda_dssim": 0.2, "densification_interval": 100, "opacity_reset_interval": 3000,
                       "densify_from_iter": 500, "densify_until_iter": 15000, "densify_grad_threshold": 0.0002},
        "model_params": {"sh_degree": 0, "white_background": False, "dynamic_model": False},
        "pipeline_params": {"convert_SHs_python": False, "compute_cov3D_python": False},
    }


def merge_config(base, override):
    out = copy.deepcopy(base)
    for k, v in (override or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge_config(out[k], v)
        else:
            out[k] = v
    return out


class SLAM:
    def __init__(self, config, dataset, save_dir=None):
        self.config = config
        self.dataset = dataset
        self.save_dir = save_dir
        ns = lambda d: types.SimpleNamespace(**d)
        self.opt_params, self.pipeline_params = ns(config["opt_params"]), ns(config["pipeline_params"])
        self.monocular = config["Dataset"]["sensor_type"] == "monocular"
        config["Training"]["monocular"] = self.monocular
        config["Results"]["save_dir"] = save_dir
        sh_degree = 3 if config["Training"]["spherical_harmonics"] else 0
        dev = dataset.device
        self.gaussians = GaussianModel(sh_degree, config=config, device=dev)
        self.gaussians.init_lr(6.0)                                            # slam.py:77
        self.gaussians.training_setup(self.opt_params)
        dynamic = config["model_params"]["dynamic_model"]
        if dynamic:
            self.gaussians.deform = DeformModel(K=3, node_num=config["Training"].get("node_num", 512), device=dev)
            self.gaussians.time_interval = 1 / max(len(dataset), 1)
        self.background = torch.tensor([1, 1, 1], dtype=torch.float32, device=dev)       # slam.py:97-98
        self.frontend, self.backend = FrontEnd(config), BackEnd(config)
        dystart = config["Training"].get("dystart", getattr(dataset, "dystart", 0) if dynamic else 0)
        for part in (self.frontend, self.backend):
            part.dataset, part.background, part.pipeline_params = dataset, self.background, self.pipeline_params
            part.dystart = dystart if dynamic else len(dataset) + 1
        self.frontend.device = str(dev)
        self.frontend.backend = self.backend
        self.frontend.set_hyperparams()
        self.backend.gaussians = self.gaussians
        self.backend.cameras_extent = 6.0                                      # slam.py:126
        self.backend.opt_params = self.opt_params
        self.backend.set_hyperparams()
        self.result = {}

    def run(self, max_frames=None, color_refinement_iters=0):
        t0 = time.perf_counter()
        self.frontend.run(max_frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fe = self.frontend
        self.gaussians = fe.gaussians
        self.result = {"frames": len(fe.cameras), "keyframes": list(fe.kf_indices), "seconds": dt, "fps": len(fe.cameras) / dt,
                       "gaussians": int(self.gaussians.get_xyz.shape[0])}
        if fe.init_done_at is not None and len(fe.cameras) > 1:          # frames per second once the map exists (tracking + keyframe mapping)
            self.result["seconds_init"] = fe.init_done_at - t0
            self.result["fps_after_init"] = (len(fe.cameras) - 1) / max(1e-9, t0 + dt - fe.init_done_at)
        if self.config["Results"].get("eval_rendering", True):
            self.result["ate_rmse"] = eval_ate(fe.cameras, fe.kf_indices, self.save_dir, 0, final=True, monocular=self.monocular)
            deltas_for = None
            if self.gaussians.deform_init:
                deltas_for = lambda frame: self.backend._deltas(frame, train=False)
            self.result["before_opt"] = eval_rendering(self._eval_frames(), self.gaussians, self.dataset, self.save_dir, self.pipeline_params,
                                                       self.background, fe.kf_indices, iteration="before_opt", deltas_for=deltas_for)
            if color_refinement_iters:
                self.backend.color_refinement(iteration_total=color_refinement_iters)
                self.result["after_opt"] = eval_rendering(self._eval_frames(), self.gaussians, self.dataset, self.save_dir, self.pipeline_params,
                                                          self.background, fe.kf_indices, iteration="after_opt", deltas_for=deltas_for)
        if self.save_dir:
            save_gaussians(self.gaussians, self.save_dir, "final", final=True)
        return self.result

    def _eval_frames(self):
        """Cameras of the tracked frames; non-keyframes were cleaned (their images dropped) but keep pose, intrinsics and time."""
        return self.frontend.cameras
