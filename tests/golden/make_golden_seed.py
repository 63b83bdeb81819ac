#!/usr/bin/env python
"""Golden vectors for the seeding of Gaussians from an RGB-D keyframe -- GaussianModel.create_pcd_from_image /
create_pcd_from_image_and_depth (gaussian_splatting/scene/gaussian_model.py:153-255) -- produced by RUNNING THE REFERENCE'S OWN METHODS
(authoring container only; /root/reference is never read at test time) under the import harness of make_golden_slam.py.

What the reference's code does itself, and what this golden therefore PINS: the exposure + clamp + byte quantisation of the image (:155-157),
the adaptive point size from the sensor depth's median (:192-194), the world-to-camera convention handed to Open3D (getWorld2View2, :203),
RGB2SH of the colours, features layout, the scale rule log(sqrt(clamp_min(dist2, 1e-7) * point_size)) with its isotropic / repeated forms,
identity rotations and inverse_sigmoid(0.5) opacities (:216-255).
What it does NOT pin, because the reference delegates it to packages that do not exist here: Open3D's RGBDImage / PointCloud calls and
simple_knn's distCUDA2. They are replaced by stand-ins that implement their DOCUMENTED behaviour -- create_from_color_and_depth(depth_scale 1,
depth_trunc 100: depths beyond the truncation become 0, colours / 255), create_from_rgbd_image(project_valid_depth_only: the pixels with
depth > 0 in row-major order, p_cam = ((u - cx) z / fx, (v - cy) z / fy, z), points = inverse(extrinsic) p_cam), random_down_sample(ratio:
int(n * ratio) points drawn without replacement -- the stand-in draws them with a seeded numpy generator and RECORDS which, the test feeds
the same pixels to the kernel), distCUDA2 = the exact mean squared distance to the 3 nearest other points (the kernel's k-NN has its own
test against oracle/knn_oracle.c).

  golden_seed.npz, per case c in (aniso, iso): c_image [3,H,W], c_depth [H,W], c_R, c_T, c_exposure (a, b), c_intr (fx, fy, cx, cy),
  c_pix [n] (v * W + u of the drawn points, output order), c_xyz, c_features_dc [n,3], c_scales, c_rots, c_opacities, c_point_size."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_slam as H  # noqa: E402  (the import harness)


class _Image:
    def __init__(self, a):
        self.a = np.asarray(a)


class _RGBD:
    @staticmethod
    def create_from_color_and_depth(rgb, depth, depth_scale=1000.0, depth_trunc=3.0, convert_rgb_to_intensity=True):
        assert depth_scale == 1.0 and not convert_rgb_to_intensity
        r = types.SimpleNamespace()
        d = depth.a.astype(np.float32) / np.float32(depth_scale)
        d = np.where(d > depth_trunc, np.float32(0), d)
        r.color, r.depth = rgb.a.astype(np.float32) / np.float32(255.0), d
        return r


class _Cloud:
    LOG = {}

    def __init__(self, pts, cols, pix):
        self.points, self.colors, self.pix = pts, cols, pix

    @staticmethod
    def create_from_rgbd_image(rgbd, intr, extrinsic=None, project_valid_depth_only=True):
        assert project_valid_depth_only
        v, u = np.nonzero(rgbd.depth > 0)                       # row-major
        z = rgbd.depth[v, u].astype(np.float64)
        pc = np.stack([(u - intr.cx) * z / intr.fx, (v - intr.cy) * z / intr.fy, z, np.ones_like(z)], 0)
        pw = (np.linalg.inv(np.asarray(extrinsic, np.float64)) @ pc).T[:, :3]
        return _Cloud(pw, rgbd.color[v, u].astype(np.float64), v * rgbd.depth.shape[1] + u)

    def random_down_sample(self, ratio):
        n = int(len(self.points) * ratio)
        keep = np.sort(_Cloud.LOG["rng"].permutation(len(self.points))[:n])        # Open3D keeps the input order of the drawn points
        _Cloud.LOG["pix"] = self.pix[keep]
        return _Cloud(self.points[keep], self.colors[keep], self.pix[keep])


def _open3d():
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace(Image=_Image, RGBDImage=_RGBD, PointCloud=_Cloud)
    o3d.camera = types.SimpleNamespace(PinholeCameraIntrinsic=lambda w, h, fx, fy, cx, cy: types.SimpleNamespace(width=w, height=h, fx=fx, fy=fy, cx=cx, cy=cy))
    return o3d


def _dist2(points):
    p = points.detach().cpu().numpy().astype(np.float64)
    d = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    return torch.tensor(np.sort(d, axis=1)[:, :3].mean(axis=1), dtype=torch.float32)


def main():
    sys.modules["open3d"] = _open3d()
    knn = types.ModuleType("simple_knn")
    knn._C = types.ModuleType("simple_knn._C")
    knn._C.distCUDA2 = _dist2
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = knn, knn._C
    sys.meta_path.insert(0, H._StubFinder())
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    with H.CudaToCpu():
        from argparse import ArgumentParser
        from arguments import ModelHiddenParams
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        hp = ModelHiddenParams(ArgumentParser())
        Hh, W = 40, 56
        fx, fy, cx, cy = 48.0, 50.0, 27.5, 20.2
        rng = np.random.default_rng(5)
        depth = rng.uniform(0.4, 3.5, (Hh, W)).astype(np.float32)
        depth[rng.uniform(size=(Hh, W)) < 0.15] = 0.0
        depth[2, 7] = 150.0                                       # beyond depth_trunc
        image = torch.tensor(rng.uniform(0, 1.15, (3, Hh, W)).astype(np.float32))
        a = 0.25
        R = torch.tensor(np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32))
        T = torch.tensor(np.array([0.15, -0.1, 0.3], np.float32))
        for case, isotropic in (("aniso", False), ("iso", True)):
            cfg = {"Dataset": {"pcd_downsample": 4, "pcd_downsample_init": 2, "point_size": 0.01, "adaptive_pointsize": True, "sensor_type": "depth"}}
            gm = GaussianModel(0, config=cfg, args=hp)
            gm.isotropic = isotropic
            cam = types.SimpleNamespace(exposure_a=torch.tensor([0.08]), exposure_b=torch.tensor([-0.04]), original_image=image, depth=depth, R=R, T=T,
                                        image_width=W, image_height=Hh, fx=fx, fy=fy, cx=cx, cy=cy, motion_mask=None)
            _Cloud.LOG["rng"] = np.random.default_rng(17)
            xyz, feats, scales, rots, opac = gm.create_pcd_from_image(cam, init=True)
            out[f"{case}_image"], out[f"{case}_depth"], out[f"{case}_R"], out[f"{case}_T"] = image.numpy(), depth, R.numpy(), T.numpy()
            out[f"{case}_exposure"], out[f"{case}_intr"] = np.array([0.08, -0.04], np.float32), np.array([fx, fy, cx, cy], np.float64)
            out[f"{case}_pix"] = _Cloud.LOG["pix"].astype(np.int64)
            out[f"{case}_xyz"], out[f"{case}_features_dc"] = xyz.numpy(), feats[:, :, 0].numpy()
            out[f"{case}_scales"], out[f"{case}_rots"], out[f"{case}_opacities"] = scales.numpy(), rots.numpy(), opac.numpy()
            out[f"{case}_point_size"] = np.float64(min(0.05, 0.01 * np.median(depth[depth > 0.1])))
            print(case, xyz.shape, feats.shape, scales.shape, float(scales.mean()))
    np.savez_compressed(os.path.join(HERE, "golden_seed.npz"), **out)


if __name__ == "__main__":
    main()
