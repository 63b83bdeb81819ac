"""GaussianModel of the SLAM loop -- the counterpart of the reference's ``gaussian_splatting/scene/gaussian_model.py`` (GM) for the
static + dynamic-subset map: same attribute names (``_xyz``, ``_features_dc``, ``_features_rest``, ``_scaling``, ``_rotation``,
``_opacity``, ``dygs``, ``unique_kfIDs``, ``n_obs``, ``max_radii2D``, ``xyz_gradient_accum``, ``denom``, ``optimizer`` with the six
named param groups), same methods (create_pcd_from_image*, extend_from_pcd*, training_setup, update_learning_rate, densify_and_prune,
prune_points, reset_opacity*, add_densification_stats, save_ply / load_ply), so ``gaussian_renderer.render`` and the loops in
slam/frontend.py, slam/backend.py use it like the reference's loops use GM.

Where the per-Gaussian work happens:
  * seeding (GM:185-255, Open3D + simple_knn in the reference) -> gsr_seed_from_rgbd (include/slam_map.h): one back-projection
    kernel, the spatial-hash 3-NN, one scale kernel; no Open3D, no host round trip of the point cloud;
  * densify_and_prune (GM:866-971: ~60 torch kernels, boolean-mask indexing of six parameters and twelve Adam moments, three
    rebuilds of the optimizer state) -> gsr_densify_select + three prefix sums + gsr_densify_apply: every tensor of the new model
    is written by ONE launch; one host read (the four totals) instead of a dozen;
  * optimizer: FusedAdam (one launch for the six groups).
The 4DGaussians HexPlane network (``_deformation``) and the SC-GS node model (``deform``) are attached by the caller when used
(deformation.deform_network / control_nodes); this class does not construct them."""
import ctypes as C
import math
import os

import numpy as np
import torch
from torch import nn

from diff_gaussian_rasterization import _C
from fused_adam import FusedAdam
import slam_losses

from . import _lib
from .ply_io import read_ply, write_ply

C0 = 0.28209479177387814


def np_median(x):
    """numpy's median of a device tensor as a host float: the mean of the two middle values for an even count (torch.median returns the
    lower one)."""
    xs = torch.sort(x.reshape(-1))[0]
    n = xs.numel()
    return float((xs[(n - 1) // 2] + xs[n // 2]) * 0.5) if n else float("nan")


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def helper(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """gaussian_splatting/utils/general_utils.py:79-94 (log-linear learning-rate schedule)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)


class GaussianModel:
    PARAM_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")     # GM:404-434, the optimizer's group order

    def __init__(self, sh_degree: int, config=None, device="cuda"):
        dev = torch.device(device)
        self.device = dev
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        E = lambda *s: torch.empty(*s, device=dev)
        M = (sh_degree + 1) ** 2
        self._xyz, self._features_dc, self._features_rest = E(0, 3), E(0, 1, 3), E(0, M - 1, 3)
        self._scaling, self._rotation, self._opacity = E(0, 3), E(0, 4), E(0, 1)
        self.max_radii2D, self.xyz_gradient_accum, self.denom = E(0), E(0, 1), E(0, 1)
        self.unique_kfIDs = torch.empty(0, dtype=torch.int32, device=dev)      # the reference keeps these two on the CPU (GM:51,55)
        self.n_obs = torch.empty(0, dtype=torch.int32, device=dev)
        self.dygs = torch.empty(0, dtype=torch.bool, device=dev)
        self.optimizer = None
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.config = config
        self.isotropic = False
        self.deform_init = False
        self.deform = None                 # SC-GS node model of the dynamic branch (slam/deform_model.DeformModel), attached by the system
        self.time_interval = 0
        self.spatial_lr_scale = 1.0
        self.generator = None              # optional torch.Generator for the split samples / down-sampling (tests pin it)

    # ---- accessors (GM:95-150) --------------------------------------------------------------------------------------
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_xyz = property(lambda s: s._xyz)
    get_dygs_xyz = property(lambda s: s._xyz.index_select(0, s.dyn_rows()))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    motion_mask = property(lambda s: torch.ones((s.dyn_rows().shape[0], 1), device=s._xyz.device, dtype=s._xyz.dtype))

    # dygs as a property: boolean-mask indexing (`_xyz[dygs]`, GM:123-125) costs a device->host synchronisation per call -- the mapping
    # loop asks for the dynamic subset several times per view -- so the row indices are kept until the mask is replaced or written.
    @property
    def dygs(self):
        return self._dygs

    @dygs.setter
    def dygs(self, value):
        self._dygs = value
        self._dyn_rows_of = None

    def dyn_rows(self):
        key = (id(self._dygs), self._dygs._version)
        if self._dyn_rows_of != key:
            self._dyn_rows_idx = self._dygs.nonzero(as_tuple=True)[0]
            self._dyn_rows_of = key
        return self._dyn_rows_idx

    def get_covariance(self, scaling_modifier=1):
        """GM:87-93,146-149 (build_scaling_rotation, strip_symmetric) -- only the compute_cov3D_python branch of render() uses it."""
        s = scaling_modifier * self.get_scaling
        s = s.repeat(1, 3) if s.shape[-1] == 1 else s
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q.unbind(-1)
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)
        L = Rm * s[:, None, :]
        cov = L @ L.transpose(1, 2)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def init_lr(self, spatial_lr_scale):
        self.spatial_lr_scale = spatial_lr_scale

    # ---- seeding (GM:151-255,321-372) ---------------------------------------------------------------------------------------
    def create_pcd_from_image(self, cam_info, init=False, scale=2.0, depthmap=None, add_dygs=False):
        """GM:151-183: pick the depth map (the caller's, the sensor's, or -- add_dygs -- the sensor depth with the static pixels
        removed) and hand over to create_pcd_from_image_and_depth."""
        cam = cam_info
        if add_dygs:
            depth = cam.depth_device().clone()
            depth[cam.motion_mask.to(depth.device)] = 0
        elif depthmap is not None:
            depth = torch.as_tensor(depthmap, dtype=torch.float32, device=self.device)
        else:
            depth = cam.depth_device()
            if self.config["Dataset"]["sensor_type"] == "monocular":
                depth = (torch.ones_like(depth) + (torch.randn(depth.shape, device=depth.device, generator=self.generator) - 0.5) * 0.05) * scale
        return self.create_pcd_from_image_and_depth(cam, cam.original_image, depth, init)

    def create_pcd_from_image_and_depth(self, cam, rgb, depth, init=False):
        """GM:185-255. `rgb` = the [3,H,W] image (exposure is applied inside the kernel, :186-188), `depth` a device [H,W] tensor."""
        ds = self.config["Dataset"]
        downsample_factor = ds["pcd_downsample_init"] if init else ds["pcd_downsample"]
        point_size = ds["point_size"]
        if ds.get("adaptive_pointsize", False):
            sensor = cam.depth_device()
            point_size = min(0.05, point_size * np_median(sensor[sensor > 0.1]))                   # :192-194 (np.median: the MEAN of the two middle values)
        depth = depth.contiguous()
        # Open3D: depth_trunc = 100 zeroes everything beyond, project_valid_depth_only keeps d > 0, random_down_sample(1 / factor)
        # keeps int(n / factor) points drawn without replacement (:195-215)
        valid = torch.nonzero(((depth > 0) & (depth <= 100.0)).view(-1), as_tuple=False).squeeze(1)
        n = int(valid.numel() * (1.0 / downsample_factor))
        perm = torch.randperm(valid.numel(), device=valid.device, generator=self.generator)[:n]
        pix = valid[perm].to(torch.int32).contiguous()
        return self.seed_from_pixels(cam, rgb, depth, pix, point_size)

    def seed_from_pixels(self, cam, rgb, depth, pix, point_size):
        """The per-point half of GM:216-255 on the device: (fused_point_cloud, features [n,3,M], scales, rots, opacities)."""
        n, dev = int(pix.numel()), self.device
        M = (self.max_sh_degree + 1) ** 2
        sd = 1 if self.isotropic else 3
        xyz = torch.empty((n, 3), device=dev)
        f_dc = torch.empty((n, 3), device=dev)
        scales, rots, opac = torch.empty((n, sd), device=dev), torch.empty((n, 4), device=dev), torch.empty((n, 1), device=dev)
        L = _lib.lib()
        ws = torch.empty((int(L.gsr_seed_workspace_size(n)),), dtype=torch.uint8, device=dev)
        rgb = rgb.to(dev, torch.float32).contiguous()
        H, W = int(depth.shape[-2]), int(depth.shape[-1])
        with torch.cuda.device(dev):
            rc = L.gsr_seed_from_rgbd(n, pix.data_ptr(), W, H, depth.data_ptr(), rgb.data_ptr(), cam.exposure_a.data_ptr(), cam.exposure_b.data_ptr(),
                                      float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), cam.R.contiguous().data_ptr(),
                                      cam.T.contiguous().data_ptr(), float(point_size), sd, xyz.data_ptr(), f_dc.data_ptr(), scales.data_ptr(),
                                      rots.data_ptr(), opac.data_ptr(), ws.data_ptr(), _lib.stream(dev))
        _lib.check(rc, "gsr_seed_from_rgbd")
        features = torch.zeros((n, 3, M), device=dev)
        features[:, :, 0] = f_dc
        return xyz, features, scales, rots, opac

    def extend_from_pcd(self, fused_point_cloud, features, scales, rots, opacities, kf_id, add_dygs=False):
        """GM:321-360."""
        n = fused_point_cloud.shape[0]
        new = {"xyz": fused_point_cloud, "f_dc": features[:, :, 0:1].transpose(1, 2).contiguous(),
               "f_rest": features[:, :, 1:].transpose(1, 2).contiguous(), "opacity": opacities, "scaling": scales, "rotation": rots}
        self.densification_postfix(new["xyz"], new["f_dc"], new["f_rest"], new["opacity"], new["scaling"], new["rotation"],
                                   torch.full((n,), bool(add_dygs), dtype=torch.bool, device=self.device),
                                   new_kf_ids=torch.full((n,), int(kf_id), dtype=torch.int32, device=self.device),
                                   new_n_obs=torch.zeros((n,), dtype=torch.int32, device=self.device))

    def extend_from_pcd_seq(self, cam_info, kf_id=-1, init=False, scale=2.0, depthmap=None, add_dygs=False):
        """GM:362-372."""
        pcd = self.create_pcd_from_image(cam_info, init, scale=scale, depthmap=depthmap, add_dygs=add_dygs)
        self.extend_from_pcd(*pcd, kf_id, add_dygs)

    # ---- optimizer (GM:400-505) ---------------------------------------------------------------------------------------
    def _params(self):
        return {"xyz": self._xyz, "f_dc": self._features_dc, "f_rest": self._features_rest, "opacity": self._opacity,
                "scaling": self._scaling, "rotation": self._rotation}

    def _set_params(self, d):
        self._xyz, self._features_dc, self._features_rest = d["xyz"], d["f_dc"], d["f_rest"]
        self._opacity, self._scaling, self._rotation = d["opacity"], d["scaling"], d["rotation"]

    def training_setup(self, training_args):
        """GM:400-447,492-505: six param groups, Adam(lr=0, eps=1e-15) -> FusedAdam with the same groups."""
        ta = training_args
        self.percent_dense = ta.percent_dense
        P = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.device)
        self.denom = torch.zeros((P, 1), device=self.device)
        for k, v in self._params().items():
            if not isinstance(v, nn.Parameter):
                self._set_params({**self._params(), k: nn.Parameter(v.requires_grad_(True))})
        lrs = {"xyz": ta.position_lr_init * self.spatial_lr_scale, "f_dc": ta.feature_lr, "f_rest": ta.feature_lr / 20.0,
               "opacity": ta.opacity_lr, "scaling": ta.scaling_lr * self.spatial_lr_scale, "rotation": ta.rotation_lr}
        groups = [{"params": [self._params()[n]], "lr": lrs[n], "name": n} for n in self.PARAM_NAMES]
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        # the backward kernels add every view's gradients to one persistent buffer (fused_adam.FusedAdam.zero_grad); Training.
        # fused_grad_accumulation = False restores autograd's own accumulation (same values)
        fused = True
        if isinstance(self.config, dict):
            fused = bool(self.config.get("Training", {}).get("fused_grad_accumulation", True))
        self.optimizer.enable_fused_gradient_accumulation(fused)
        self.lr_init = ta.position_lr_init * self.spatial_lr_scale
        self.lr_final = ta.position_lr_final * self.spatial_lr_scale
        self.lr_delay_mult = ta.position_lr_delay_mult
        self.max_steps = ta.position_lr_max_steps

    def xyz_lr_at(self, iteration):
        """The position learning rate update_learning_rate(iteration) sets (GM:492-505)."""
        return helper(iteration, lr_init=self.lr_init, lr_final=self.lr_final, lr_delay_mult=self.lr_delay_mult, max_steps=self.max_steps)

    def update_learning_rate(self, iteration):
        """GM:492-505."""
        lr = None
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                lr = self.xyz_lr_at(iteration)
                group["lr"] = lr
        return lr

    # ---- PLY I/O (GM:567-620,640-733) ---------------------------------------------------------------------------------------
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        names.append("dygs")
        return names

    def save_ply(self, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy(),
                self.dygs.detach().cpu().numpy().reshape(-1, 1).astype(np.float32)]
        write_ply(path, self.construct_list_of_attributes(), np.concatenate(cols, axis=1).astype(np.float32))

    def load_ply(self, path):
        names, data = read_ply(path)
        col = {n: data[:, i] for i, n in enumerate(names)}
        P = data.shape[0]
        dev = self.device
        T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        xyz = np.stack([col["x"], col["y"], col["z"]], axis=1)
        if "dygs" in col:
            self.dygs = torch.tensor(col["dygs"] != 0, dtype=torch.bool, device=dev)
        else:
            self.dygs = torch.zeros(P, dtype=torch.bool, device=dev)
        f_dc = np.stack([col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]], axis=1)[:, :, None]                 # [P,3,1]
        extra = sorted((n for n in names if n.startswith("f_rest_")), key=lambda x: int(x.split("_")[-1]))
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_extra = np.stack([col[n] for n in extra], axis=1).reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1) if extra else np.zeros((P, 3, 0), np.float32)
        sc = sorted((n for n in names if n.startswith("scale_")), key=lambda x: int(x.split("_")[-1]))
        rt = sorted((n for n in names if n.startswith("rot_")), key=lambda x: int(x.split("_")[-1]))
        self._xyz = nn.Parameter(T(xyz).requires_grad_(True))
        self._features_dc = nn.Parameter(T(f_dc).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(T(f_extra).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(T(col["opacity"][:, None]).requires_grad_(True))
        self._scaling = nn.Parameter(T(np.stack([col[n] for n in sc], axis=1)).requires_grad_(True))
        self._rotation = nn.Parameter(T(np.stack([col[n] for n in rt], axis=1)).requires_grad_(True))
        self.isotropic = len(sc) == 1
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros((P,), device=dev)
        self.unique_kfIDs = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.n_obs = torch.zeros((P,), dtype=torch.int32, device=dev)
        # densification statistics and the optimizer belong to the tensors that were just replaced: fresh statistics, and the optimizer
        # (if one was set up) is dropped -- call training_setup() again before training the loaded map
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.optimizer = None

    # ---- optimizer-state surgery (GM:734-865) -----------------------------------------------------------------------------
    def _group(self, name):
        for g in self.optimizer.param_groups:
            if g.get("name") == name:
                return g
        raise KeyError(name)

    def replace_tensor_to_optimizer(self, tensor, name):
        """GM:734-748."""
        out = {}
        g = self._group(name)
        st = self.optimizer.state.get(g["params"][0], None)
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
            del self.optimizer.state[g["params"][0]]
        g["params"][0] = nn.Parameter(tensor.requires_grad_(True))
        if st is not None:
            self.optimizer.state[g["params"][0]] = st
        out[name] = g["params"][0]
        return out

    def reset_opacity(self):
        """GM:622-625."""
        new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.01)
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def reset_opacity_nonvisible(self, visibility_filters):
        """GM:627-636."""
        new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.4)
        cur = self.get_opacity
        for f in visibility_filters:
            new[f] = cur[f]
        self._opacity = self.replace_tensor_to_optimizer(new.detach(), "opacity")["opacity"]

    def _rebuild(self, flags, offsets, n_keep, n_clone, n_split, n_child, noise):
        """Write the whole post-densification model (six parameters, their twelve Adam moments, dygs / kf ids / n_obs) with ONE
        gsr_densify_apply launch and re-register the parameters with the optimizer (GM:750-865 does this tensor by tensor)."""
        dev, P = self.device, int(self._xyz.shape[0])
        n_out = n_keep + n_clone + 2 * n_child
        params = self._params()
        desc, keep, new_params, new_states = [], [], {}, {}
        kinds = {"xyz": _lib.XYZ, "scaling": _lib.SCALE}

        def add(src, kind):
            src = src.detach().contiguous()
            width = int(src.numel() // max(P, 1)) if P else int(np.prod(src.shape[1:]))
            dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
            keep.append(src)
            desc.append((src.data_ptr(), dst.data_ptr(), max(width, 1), kind))
            return dst

        for name in self.PARAM_NAMES:
            p = params[name]
            if p.numel() == 0 and p.shape[0] == P and int(np.prod(p.shape[1:])) == 0:       # f_rest of an SH-degree-0 model: [P,0,3]
                new_params[name] = torch.empty((n_out,) + tuple(p.shape[1:]), device=dev)
                st = self.optimizer.state.get(p, None) if self.optimizer is not None else None
                if st is not None and "exp_avg" in st:
                    new_states[name] = (torch.empty_like(new_params[name]), torch.empty_like(new_params[name]))
                continue
            new_params[name] = add(p, kinds.get(name, _lib.COPY))
            st = self.optimizer.state.get(p, None) if self.optimizer is not None else None
            if st is not None and "exp_avg" in st:
                new_states[name] = (add(st["exp_avg"], _lib.STATE), add(st["exp_avg_sq"], _lib.STATE))
        aux = {"dygs": self.dygs.to(torch.int32), "kf": self.unique_kfIDs.to(torch.int32), "n_obs": self.n_obs.to(torch.int32)}
        new_aux = {k: add(v, _lib.COPY) for k, v in aux.items()}
        arr = (_lib.DensifyTensor * len(desc))()
        for k, (s_, d_, w_, kind) in enumerate(desc):
            arr[k].src, arr[k].dst, arr[k].width, arr[k].kind = s_, d_, w_, kind
        L = _lib.lib()
        sd = int(self._scaling.shape[1])
        with torch.cuda.device(dev):
            rc = L.gsr_densify_apply(P, flags.data_ptr(), offsets.data_ptr(), n_keep, n_clone, n_split, n_child, len(desc), arr,
                                     self._xyz.detach().contiguous().data_ptr(), self._scaling.detach().contiguous().data_ptr(), sd,
                                     self._rotation.detach().contiguous().data_ptr(), noise.data_ptr() if noise is not None and noise.numel() else None,
                                     _lib.stream(dev))
        _lib.check(rc, "gsr_densify_apply")
        # re-register with the optimizer (what _prune_optimizer / cat_tensors_to_optimizer do, GM:750-840)
        out = {}
        for name in self.PARAM_NAMES:
            newp = nn.Parameter(new_params[name].requires_grad_(True))
            if self.optimizer is not None:
                g = self._group(name)
                old = g["params"][0]
                st = self.optimizer.state.pop(old, None)
                g["params"][0] = newp
                if st is not None:
                    if name in new_states:
                        st["exp_avg"], st["exp_avg_sq"] = new_states[name]
                    self.optimizer.state[newp] = st
            out[name] = newp
        self._set_params(out)
        self.dygs = new_aux["dygs"].to(torch.bool)
        self.unique_kfIDs, self.n_obs = new_aux["kf"], new_aux["n_obs"]

    def prune_points(self, mask):
        """GM:786-810: drop the Gaussians selected by the boolean `mask`."""
        P = int(self._xyz.shape[0])
        if P == 0:
            return
        keep = (~mask.to(self.device).bool()).to(torch.int32)
        flags = torch.zeros((4, P), dtype=torch.int32, device=self.device)
        flags[0] = keep
        offsets = torch.zeros_like(flags)
        offsets[0] = torch.cumsum(keep, 0, dtype=torch.int32) - keep
        n_keep = int(keep.sum().item())
        keepb = keep.bool()
        acc, den, rad = self.xyz_gradient_accum[keepb], self.denom[keepb], self.max_radii2D[keepb]
        self._rebuild(flags, offsets, n_keep, 0, 0, 0, None)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = acc, den, rad

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation, new_dygs,
                              new_kf_ids=None, new_n_obs=None):
        """GM:842-864: append rows (used by the seeding path; densify_and_prune writes its result in one launch instead)."""
        new = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities, "scaling": new_scaling,
               "rotation": new_rotation}
        out = {}
        for name in self.PARAM_NAMES:
            old = self._params()[name]
            ext = new[name].detach().to(self.device)
            cat = nn.Parameter((torch.cat((old.detach(), ext), dim=0) if old.shape[0] else ext.clone()).requires_grad_(True))
            if self.optimizer is not None:
                g = self._group(name)
                st = self.optimizer.state.pop(g["params"][0], None)
                if st is not None and "exp_avg" in st:
                    st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                    st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                g["params"][0] = cat
                if st is not None:
                    self.optimizer.state[cat] = st
            out[name] = cat
        self._set_params(out)
        P = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.device)
        self.denom = torch.zeros((P, 1), device=self.device)
        self.max_radii2D = torch.zeros((P,), device=self.device)
        self.dygs = torch.cat((self.dygs, new_dygs.to(self.device)))
        if new_kf_ids is not None:
            self.unique_kfIDs = torch.cat((self.unique_kfIDs, new_kf_ids.to(self.device, torch.int32))).int()
        if new_n_obs is not None:
            self.n_obs = torch.cat((self.n_obs, new_n_obs.to(self.device, torch.int32))).int()

    # ---- densification (GM:866-977) -------------------------------------------------------------------------------------
    def densify_select(self, max_grad, min_opacity, extent, max_screen_size):
        """int32 [4,P] decisions of clone / split / prune (include/slam_map.h, gsr_densify_select)."""
        P, dev = int(self._xyz.shape[0]), self.device
        flags = torch.empty((4, P), dtype=torch.int32, device=dev)
        L = _lib.lib()
        with torch.cuda.device(dev):
            rc = L.gsr_densify_select(P, self.xyz_gradient_accum.contiguous().data_ptr(), self.denom.contiguous().data_ptr(),
                                      self._scaling.detach().contiguous().data_ptr(), int(self._scaling.shape[1]),
                                      self._opacity.detach().contiguous().data_ptr(), float(max_grad), float(self.percent_dense * extent),
                                      float(min_opacity), float(0.1 * extent) if max_screen_size else -1.0, flags.data_ptr(), _lib.stream(dev))
        _lib.check(rc, "gsr_densify_select")
        return flags

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, noise=None):
        """GM:953-971 (densify_and_clone :920-951, densify_and_split :866-918, prune) as select -> scan -> apply. `noise`
        ([2 n_split, 3] standard normal, the reference's torch.normal draw) can be passed in to pin the result."""
        P = int(self._xyz.shape[0])
        if P == 0:
            return
        flags = self.densify_select(max_grad, min_opacity, extent, max_screen_size)
        incl = torch.cumsum(flags, dim=1, dtype=torch.int32)
        offsets = (incl - flags).contiguous()
        n_keep, n_clone, n_split, n_child = (int(v) for v in incl[:, -1].tolist())          # the one host read
        if noise is None:
            noise = torch.randn((2 * n_split, 3), device=self.device, generator=self.generator)
        self._rebuild(flags, offsets, n_keep, n_clone, n_split, n_child, noise.to(self.device, torch.float32).contiguous())
        n_out = int(self._xyz.shape[0])
        self.xyz_gradient_accum = torch.zeros((n_out, 1), device=self.device)               # GM:855-857
        self.denom = torch.zeros((n_out, 1), device=self.device)
        self.max_radii2D = torch.zeros((n_out,), device=self.device)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """GM:973-977 with the reference's signature (boolean filter); the loops call slam_losses.add_densification_stats(self, pts, radii),
        which also takes the max of the screen radii, in one launch."""
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def add_view_stats(self, viewspace_point_tensor, radii):
        """utils/slam_backend.py:712-720 for one rendered view, fused (gsr_densification_stats)."""
        acc, den = self.xyz_gradient_accum.view(-1), self.denom.view(-1)
        holder = type("S", (), {})()
        holder.max_radii2D, holder.xyz_gradient_accum, holder.denom = self.max_radii2D, acc, den
        slam_losses.add_densification_stats(holder, viewspace_point_tensor, radii)
