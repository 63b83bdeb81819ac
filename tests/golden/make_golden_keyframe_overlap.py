#!/usr/bin/env python
"""Golden vectors for Camera.keyframe_selection_overlap (utils/camera_utils.py:319-365, with get_pointcloud :236-265), produced by
IMPORTING and RUNNING the reference's own method (authoring container only; /root/reference is never read at test time) under the
harness of make_golden_slam.py (device='cuda' -> 'cpu' rewrite, stand-in modules for RAFT / GMA / open3d ...).

  golden_keyframe_overlap.npz, per case c in (a, b):
    c_depth [H,W], c_intr (fx, fy, cx, cy, W, H), c_ids [K], c_R [K,3,3], c_T [K,3]   the keyframes (the newest one is c_self)
    c_self, c_time, c_seed                                                              the call: viewpoints[c_self].keyframe_selection_overlap(dataset, viewpoints, c_time)
                                                                                        after np.random.seed(c_seed)
    c_selected                                                                          what it returned
    c_sorted                                                                            the candidate ids in the order the permutation saw them (by
                                                                                        percent_inside, descending; recorded with an identity permutation)
  case a: generic poses, depth with holes; case b: the newest keyframe at the identity pose with a centred principal point -- mirrored
  pixels then produce equal |coordinates| after rounding and get_pointcloud's duplicate filter removes BOTH of every such pair."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_slam as H  # noqa: E402  (the import harness)


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    M = {"x": [[1, 0, 0], [0, c, -s], [0, s, c]], "y": [[c, 0, s], [0, 1, 0], [-s, 0, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    return np.array(M, np.float32)


def main():
    sys.meta_path.insert(0, H._StubFinder())
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    with H.CudaToCpu():
        from utils.camera_utils import Camera
        for case, (W, Hh, cx, cy, seed, identity_self) in {"a": (64, 48, 31.3, 24.6, 11, False), "b": (64, 48, 32.0, 24.0, 5, True)}.items():
            rng = np.random.default_rng(seed)
            fx = fy = 55.0
            yy, xx = np.mgrid[0:Hh, 0:W]
            depth = (2.0 + 0.6 * np.sin(xx / 9.0) + 0.4 * np.cos(yy / 7.0)).astype(np.float32)
            if identity_self:
                depth = np.full((Hh, W), 2.5, np.float32)           # constant depth: mirrored pixels give equal |x|, |y|, z
            depth[rng.uniform(size=depth.shape) < 0.1] = 0.0       # holes
            ids = [0, 2, 5, 7, 9, 11, 13, 14, 16, 19, 23]
            cams = {}
            for k, i in enumerate(ids):
                # a trajectory that turns away from the newest view: the older the keyframe, the smaller the overlap; one looks backwards
                age = len(ids) - 1 - k
                R = rot("y", 0.055 * age) @ rot("x", 0.02 * age)
                if i == 5:
                    R = rot("y", 3.0)                                # looks the other way: percent_inside == 0 -> never selected
                T = np.array([0.05 * age, -0.02 * age, 0.03 * age], np.float32)
                if identity_self and i == ids[-1]:
                    R, T = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
                cams[i] = types.SimpleNamespace(uid=i, R=torch.tensor(R), T=torch.tensor(T))
            me = cams[ids[-1]]
            me.depth, me.device = depth, "cpu"
            me.get_pointcloud = lambda *a, me=me: Camera.get_pointcloud(me, *a)
            dataset = types.SimpleNamespace(fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=Hh)
            time = ids[-3]
            np.random.seed(seed)
            selected = Camera.keyframe_selection_overlap(me, dataset, cams, time)
            keep = np.random.permutation
            np.random.permutation = lambda a: a
            try:
                sorted_ids = Camera.keyframe_selection_overlap(me, dataset, cams, time, pose_window=-100)
            finally:
                np.random.permutation = keep
            out[f"{case}_depth"], out[f"{case}_intr"] = depth, np.array([fx, fy, cx, cy, W, Hh], np.float64)
            out[f"{case}_ids"] = np.array(ids, np.int64)
            out[f"{case}_R"] = np.stack([cams[i].R.numpy() for i in ids])
            out[f"{case}_T"] = np.stack([cams[i].T.numpy() for i in ids])
            out[f"{case}_self"], out[f"{case}_time"], out[f"{case}_seed"] = np.int64(ids[-1]), np.int64(time), np.int64(seed)
            out[f"{case}_selected"] = np.array([int(v) for v in selected], np.int64)
            out[f"{case}_sorted"] = np.array([int(v) for v in sorted_ids], np.int64)
            print(case, "selected", out[f"{case}_selected"], "sorted", out[f"{case}_sorted"])
    np.savez_compressed(os.path.join(HERE, "golden_keyframe_overlap.npz"), **out)


if __name__ == "__main__":
    main()
