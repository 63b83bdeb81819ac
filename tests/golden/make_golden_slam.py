#!/usr/bin/env python
"""Golden vectors for the SLAM-loop components, produced by IMPORTING and RUNNING the reference's own Python
(authoring container only; /root/reference is never read at test time). Everything runs on CPU under a device='cuda' -> 'cpu'
rewrite; external packages the reference imports but this path never executes (open3d, plyfile, pytorch3d, evo, wandb, cv2,
torchvision, ultralytics, RAFT, GMA, the GUI) are replaced by empty stand-in modules.

  golden_slam.npz:
    densify_*      GaussianModel.densify_and_prune (gaussian_splatting/scene/gaussian_model.py:953-971) on a seeded 600-Gaussian model
                   with live Adam moments: every parameter, both moments of every group, dygs / unique_kfIDs / n_obs after the call,
                   and the normal samples torch.normal drew inside densify_and_split (recorded by wrapping torch.normal);
                   two cases: anisotropic + size pruning, isotropic without.
    prune_*        GaussianModel.prune_points (:786-810) on the same model.
    window_*       FrontEnd.is_keyframe / add_to_window (utils/slam_frontend.py:472-562) on seeded poses and visibility sets.
    align_*        utils/eval_utils.py align / evaluate_ate on a noisy rigidly transformed trajectory.
    pose_*         utils/pose_utils.py update_pose for small and large tau.
    gradmask_*     Camera.compute_grad_mask (utils/camera_utils.py:205-233), psnr (utils/image_utils.py), get_median_depth, lr helper.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)

STUB_ROOTS = {"open3d", "plyfile", "simple_knn", "pytorch3d", "tkinter", "cv2", "evo", "wandb", "torchmetrics", "torchvision", "ultralytics",
              "RAFT", "GMA", "flow_utils", "gui", "munch", "glfw", "OpenGL", "imgviz", "lpips", "trimesh", "rich", "diff_gaussian_rasterization"}


class _Any(types.ModuleType):
    __all__ = []
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Any(spec.name)

    def exec_module(self, module):
        pass


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        d = kwargs.get("device")
        if d is not None and "cuda" in str(d):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def main():
    sys.meta_path.insert(0, _StubFinder())
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    with CudaToCpu():
        from argparse import ArgumentParser
        from arguments import ModelHiddenParams
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from gaussian_splatting.utils.general_utils import helper
        from gaussian_splatting.utils.image_utils import psnr
        import utils.slam_utils as su
        import utils.pose_utils as pu
        import utils.eval_utils as eu

        hp = ModelHiddenParams(ArgumentParser())
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                                    position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)

        def build(P, isotropic, seed):
            g = torch.Generator().manual_seed(seed)
            R = lambda *s: torch.randn(*s, generator=g)
            gm = GaussianModel(0, config={"Dataset": {}}, args=hp)
            gm.init_lr(6.0)
            gm.training_setup(opt)
            feats = torch.zeros(P, 3, 1)
            feats[:, :, 0] = R(P, 3)
            scales = (R(P, 1) * 0.8 - 2.6) if isotropic else (R(P, 3) * 0.8 - 2.6)
            gm.isotropic = isotropic
            gm.extend_from_pcd(R(P, 3), feats, scales, R(P, 4), R(P, 1) * 2.0, kf_id=3, add_dygs=False)
            gm.dygs = torch.rand(P, generator=g) < 0.2
            gm.unique_kfIDs = torch.randint(0, 9, (P,), generator=g).int()
            gm.n_obs = torch.randint(0, 5, (P,), generator=g).int()
            # two Adam steps so that the moments are live
            for _ in range(2):
                for grp in gm.optimizer.param_groups:
                    p = grp["params"][0]
                    p.grad = torch.randn(p.shape, generator=g) * 0.01
                gm.optimizer.step()
                gm.optimizer.zero_grad(set_to_none=True)
            gm.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 0.001
            gm.denom = torch.randint(0, 4, (P, 1), generator=g).float()          # zeros -> NaN grads -> 0
            gm.max_radii2D = torch.rand(P, generator=g) * 40
            return gm

        def snapshot(gm, tag):
            d = {}
            names = {"xyz": gm._xyz, "f_dc": gm._features_dc, "f_rest": gm._features_rest, "opacity": gm._opacity, "scaling": gm._scaling,
                     "rotation": gm._rotation}
            for n, p in names.items():
                d[f"{tag}_{n}"] = p.detach().numpy().copy()
                st = gm.optimizer.state.get(p, None)
                if st is not None and "exp_avg" in st:
                    d[f"{tag}_{n}_m"] = st["exp_avg"].numpy().copy()
                    d[f"{tag}_{n}_v"] = st["exp_avg_sq"].numpy().copy()
            d[f"{tag}_dygs"] = gm.dygs.numpy().copy()
            d[f"{tag}_kf"] = gm.unique_kfIDs.numpy().copy()
            d[f"{tag}_nobs"] = gm.n_obs.numpy().copy()
            d[f"{tag}_accum"] = gm.xyz_gradient_accum.numpy().copy()
            d[f"{tag}_denom"] = gm.denom.numpy().copy()
            d[f"{tag}_radii"] = gm.max_radii2D.numpy().copy()
            return d

        for case, (iso, extent, screen) in {"a": (False, 6.0, 20), "b": (True, 3.0, None)}.items():
            gm = build(600, iso, 11 if case == "a" else 12)
            out.update(snapshot(gm, f"densify_{case}_in"))
            rec = {}
            real_normal = torch.normal

            def normal(*a, **k):
                # the standard-normal draw behind torch.normal(mean, std): same generator state, same shape (checked below)
                state = torch.get_rng_state()
                r = real_normal(*a, **k)
                std, mean = k.get("std"), k.get("mean")
                after = torch.get_rng_state()
                torch.set_rng_state(state)
                z = torch.randn(r.shape)
                torch.set_rng_state(after)
                assert torch.equal(z * std + mean, r.detach()), "torch.normal(mean, std) != mean + std * randn for this build"
                rec["noise"] = z.numpy().copy()
                return r

            torch.normal = normal
            torch.manual_seed(5)
            args = dict(max_grad=0.0002, min_opacity=0.3 if case == "a" else 0.05, extent=extent, max_screen_size=screen)
            with torch.no_grad():                                    # as utils/slam_backend.py:257,711 call it
                gm.densify_and_prune(args["max_grad"], args["min_opacity"], args["extent"], args["max_screen_size"])
            torch.normal = real_normal
            out.update(snapshot(gm, f"densify_{case}_out"))
            out[f"densify_{case}_noise"] = rec.get("noise", np.zeros((0, 3), np.float32))
            out[f"densify_{case}_args"] = np.array([args["max_grad"], args["min_opacity"], args["extent"], -1 if screen is None else screen, 0.01], np.float64)

        gm = build(300, False, 21)
        out.update(snapshot(gm, "prune_in"))
        mask = torch.rand(300, generator=torch.Generator().manual_seed(3)) < 0.35
        with torch.no_grad():
            gm.prune_points(mask)
        out.update(snapshot(gm, "prune_out"))
        out["prune_mask"] = mask.numpy()

        # ---- keyframe management ----
        from utils.slam_frontend import FrontEnd
        cfg = {"Training": {"monocular": False, "kf_translation": 0.08, "kf_min_translation": 0.05, "kf_overlap": 0.9, "kf_cutoff": 0.3,
                            "window_size": 8}, "model_params": {"dynamic_model": False}}
        fe = FrontEnd(cfg)
        g = torch.Generator().manual_seed(4)
        cams = {}
        for k in range(12):
            a = 0.03 * k
            Rm = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
            cams[k] = types.SimpleNamespace(R=Rm, T=torch.tensor([0.05 * k, 0.01 * k * (k % 3), 0.02 * k]), uid=k)
        fe.cameras = cams
        P = 500
        vis = {k: (torch.rand(P, generator=g) < (0.9 - 0.06 * k)).long() for k in range(12)}
        out["window_R"] = np.stack([cams[k].R.numpy() for k in range(12)])
        out["window_T"] = np.stack([cams[k].T.numpy() for k in range(12)])
        out["window_vis"] = np.stack([vis[k].numpy() for k in range(12)])
        res_kf = []
        for (cur, last, med) in [(5, 3, 2.0), (6, 5, 1.0), (9, 2, 3.0), (11, 10, 0.5), (7, 6, 5.0)]:
            fe.median_depth = med
            res_kf.append([cur, last, med, float(bool(fe.is_keyframe(cur, last, vis[cur], vis)))])
        out["window_is_keyframe"] = np.array(res_kf)
        res_w = []
        for init_flag, window, cur in [(True, [8, 7, 6, 5, 4, 3, 2, 1], 9), (True, [5, 4, 3], 6), (False, [10, 9, 8, 7, 6, 5, 4, 3], 11),
                                       (True, [10, 8, 6, 4, 2, 1, 0, 3], 11)]:
            fe.initialized = init_flag
            w, removed = fe.add_to_window(cur, vis[cur], vis, list(window))
            res_w.append(np.array([int(init_flag), cur, -1 if removed is None else removed, len(window)] + list(window) + [-9] + list(w) + [-9] * (12 - len(w))))
        out["window_add"] = np.stack([r[:32] if len(r) >= 32 else np.pad(r, (0, 32 - len(r)), constant_values=-9) for r in res_w])

        # ---- trajectory alignment ----
        g = torch.Generator().manual_seed(8)
        n = 40
        gt = torch.cumsum(torch.randn(n, 3, generator=g) * 0.05, 0).numpy().astype(np.float64)
        ang = 0.4
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        est = (Rz @ gt.T).T + np.array([0.3, -0.2, 0.1]) + torch.randn(n, 3, generator=g).numpy() * 0.01
        rot, trans, err = eu.align(np.asmatrix(gt.T), np.asmatrix(est.T))
        mk = lambda p: [np.block([[np.eye(3), q[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]) for q in p]
        out["align_gt"], out["align_est"] = gt, est
        out["align_rot"], out["align_trans"], out["align_err"] = np.asarray(rot), np.asarray(trans), np.asarray(err)
        out["align_ate_mean"] = np.array(eu.evaluate_ate(mk(gt), mk(est)))

        # ---- pose update ----
        res = []
        for tau in ([1e-3, -2e-3, 5e-4, 2e-3, -1e-3, 3e-3], [0.2, -0.1, 0.3, 0.5, -0.7, 0.2], [1e-7, 0, 0, 1e-7, 0, 1e-7]):
            cam = types.SimpleNamespace(R=torch.tensor([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]]), T=torch.tensor([0.1, -0.2, 0.3]),
                                        cam_trans_delta=torch.tensor(tau[:3], dtype=torch.float32), cam_rot_delta=torch.tensor(tau[3:], dtype=torch.float32))
            cam.update_RT = lambda R, t, c=cam: (setattr(c, "R", R), setattr(c, "T", t))
            conv = pu.update_pose(cam)
            res.append(np.concatenate([np.array(tau), cam.R.numpy().ravel(), cam.T.numpy(), [float(bool(conv))]]))
        out["pose_cases"] = np.stack(res)

        # ---- small helpers ----
        g = torch.Generator().manual_seed(9)
        img = torch.rand(3, 48, 64, generator=g)
        img[:, :6, :10] = 0.0
        from utils.camera_utils import Camera
        fake = types.SimpleNamespace(original_image=img, grad_mask=None)
        Camera.compute_grad_mask(fake, {"Training": {"edge_threshold": 1.1}, "Dataset": {"type": "tum"}})
        out["gradmask_image"], out["gradmask_mask"] = img.numpy(), fake.grad_mask.numpy()
        a, b = torch.rand(1, 500, generator=g), torch.rand(1, 500, generator=g)
        out["psnr_in"], out["psnr_out"] = np.stack([a.numpy(), b.numpy()]), psnr(a, b).numpy()
        depth = torch.rand(1, 48, 64, generator=g) * 4
        opac = torch.rand(1, 48, 64, generator=g)
        out["median_in"] = np.stack([depth.numpy(), opac.numpy()])
        out["median_out"] = np.array(float(su.get_median_depth(depth, opac)))
        out["lr_helper"] = np.array([[s, helper(s, lr_init=0.00096, lr_final=0.0000096, lr_delay_mult=0.01, max_steps=30000)] for s in (0, 1, 100, 5000, 30000, 40000)])

    path = os.path.join(HERE, "golden_slam.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in list(out.items())[:8]}, "...", len(out), "arrays")


if __name__ == "__main__":
    main()
