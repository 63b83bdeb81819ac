"""world_size-2 gloo test of the view-sharded mapping BACK-END: ``slam.backend.BackEnd.map_static()`` itself runs on two ranks -- window
keyframes and random keyframes rendered by their owners, owner-local pose steps, ONE gradient all-reduce per iteration, the statistics /
visibility reductions on the iterations that use them -- and must end with the map, the camera poses and the covisibility rows a single
process produces from the same state (utils/slam_backend.py:1013-1224 is the loop being sharded).

There is no GPU here, so the pieces of the back-end that are HIP kernels are replaced by CPU stand-ins, all of them test doubles defined
in this file: the renderer is the C oracle behind oracle/torch_binding.py, the mapping loss a plain L1, the camera step a gradient step,
the model a five-tensor toy with torch.optim.Adam. Everything that is being tested -- the control flow of map_static(), ViewShard's
ownership and collectives, the order of reductions and optimizer steps -- is the product's."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import REPO, PKG, make_camera, make_gaussians, keyframe_pose

W, H, P = 64, 48, 300


class ToyCamera:
    def __init__(self, uid):
        R, t = keyframe_pose(uid)
        self.uid, self.R, self.T = uid, torch.tensor(np.asarray(R, np.float32)), torch.tensor(np.asarray(t, np.float32))
        self.cam_rot_delta = torch.nn.Parameter(torch.zeros(3))
        self.cam_trans_delta = torch.nn.Parameter(torch.zeros(3))
        self.exposure_a = torch.nn.Parameter(torch.zeros(1))
        self.exposure_b = torch.nn.Parameter(torch.zeros(1))
        rng = np.random.default_rng(100 + uid)
        self.gt_image = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
        self.gt_depth = torch.tensor(rng.uniform(1, 4, (1, H, W)).astype(np.float32))
        self.steps = 0

    def update_RT(self, R, T):
        self.R, self.T = R.clone(), T.clone()

    def pose_step(self, lr_rot, lr_trans, lr_exp, optimize_pose=True, optimize_exposure=True, latch=False):
        with torch.no_grad():      # a plain gradient step on the translation and the exposure: enough to see who stepped what
            if optimize_pose and self.cam_trans_delta.grad is not None:
                self.T = self.T - 50.0 * lr_trans * self.cam_trans_delta.grad.reshape(3)
            if optimize_exposure and self.exposure_a.grad is not None:
                self.exposure_a -= lr_exp * self.exposure_a.grad
                self.exposure_b -= lr_exp * self.exposure_b.grad
        for p in (self.cam_rot_delta, self.cam_trans_delta, self.exposure_a, self.exposure_b):
            p.grad = None
        self.steps += 1


class ToyModel:
    deform_init = False

    def __init__(self):
        g = make_gaussians(P, make_camera(W, H), seed=3, scale_mean=0.04)
        T = lambda a: torch.nn.Parameter(torch.tensor(np.asarray(a, np.float32)))
        self._xyz, self._shs, self._opacity = T(g["means3D"]), T(g["shs"]), T(g["opacities"])
        self._scaling, self._rotation = T(np.log(g["scales"])), T(g["rotations"])
        self.optimizer = torch.optim.Adam([{"params": [p], "lr": lr, "name": n} for p, lr, n in (
            (self._xyz, 1e-3, "xyz"), (self._shs, 2e-3, "f_dc"), (self._opacity, 5e-3, "opacity"), (self._scaling, 1e-3, "scaling"),
            (self._rotation, 1e-3, "rotation"))], eps=1e-15)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)
        self.n_obs = torch.zeros(P, dtype=torch.int32)
        self.events = []

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))

    def add_view_stats(self, viewspace_points, radii):
        vis = radii > 0
        self.xyz_gradient_accum[vis] += viewspace_points.grad[vis, :2].norm(dim=-1, keepdim=True)
        self.denom[vis] += 1
        self.max_radii2D[vis] = torch.maximum(self.max_radii2D[vis], radii[vis].float())

    def densify_and_prune(self, *a):          # keeps the row count: records what it was given, then resets the statistics
        self.events.append(("densify", self.xyz_gradient_accum.clone(), self.denom.clone(), self.max_radii2D.clone()))
        self.xyz_gradient_accum.zero_(); self.denom.zero_(); self.max_radii2D.zero_()

    def reset_opacity_nonvisible(self, filters):
        self.events.append(("reset", torch.stack(list(filters)).any(dim=0).clone()))

    def update_learning_rate(self, it):
        pass


def toy_render(self, viewpoint, deltas):
    import oracle.torch_binding as ob
    cam = make_camera(W, H, R=viewpoint.R.numpy().astype(np.float64), t=viewpoint.T.numpy().astype(np.float64))
    T = lambda a: torch.tensor(np.asarray(a, np.float32))
    rs = ob.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.ones(3), 1.0, T(cam.viewmatrix), T(cam.projmatrix),
                                          T(cam.projmatrix_raw), 0, T(cam.campos), False, False)
    g = self.gaussians
    pts = torch.zeros(P, 3, requires_grad=True)
    rot = torch.nn.functional.normalize(g._rotation)
    c, radii, d, o, n = ob.GaussianRasterizer(rs)(means3D=g._xyz, means2D=pts, opacities=g._opacity, shs=g._shs, scales=torch.exp(g._scaling),
                                                  rotations=rot, theta=viewpoint.cam_rot_delta, rho=viewpoint.cam_trans_delta)
    return {"render": c, "viewspace_points": pts, "visibility_filter": radii > 0, "radii": radii, "depth": d, "opacity": o, "n_touched": n}


def toy_loss(config, image, depth, viewpoint, opacity, **kw):
    img = torch.exp(viewpoint.exposure_a) * image + viewpoint.exposure_b
    return 0.9 * (img - viewpoint.gt_image).abs().mean() + 0.1 * (depth - viewpoint.gt_depth).abs().mean()


def run_backend(iters, window, all_kfs):
    """Builds the toy state, runs map_static(iters) + the pruning call, returns everything that must agree across worlds."""
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from slam import backend as be
    torch.manual_seed(7)                                 # every rank alike: the random keyframes of an iteration are the same draw
    cfg = {"Training": {"pose_window": 3, "window_size": len(window), "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}},
           "model_params": {"dynamic_model": False}}
    b = be.BackEnd(cfg)
    b.device = "cpu"
    b.gaussians = ToyModel()
    b.opt_params = types.SimpleNamespace(densify_grad_threshold=0.0002)
    b.loss_values, b.gaussian_update_every, b.gaussian_update_offset, b.gaussian_reset = False, 3, 2, 4
    b.gaussian_th = b.gaussian_extent = b.size_threshold = 0.0
    b.viewpoints = {k: ToyCamera(k) for k in all_kfs}
    b._render = types.MethodType(toy_render, b)
    b._render_many = lambda cams, deltas: [toy_render(b, c, d) for c, d in zip(cams, deltas)]       # (the product's is one multi-view launch chain)
    be.slam_losses = types.SimpleNamespace(get_loss_mapping=toy_loss)
    b.map_static(window, iters=iters)
    b.map_static(window, prune=True)
    g = b.gaussians
    return {"params": [p.detach().numpy().copy() for p in (g._xyz, g._shs, g._opacity, g._scaling, g._rotation)],
            "poses": {k: (v.R.numpy().copy(), v.T.numpy().copy(), v.exposure_a.item(), v.exposure_b.item()) for k, v in b.viewpoints.items()},
            "steps": {k: v.steps for k, v in b.viewpoints.items()},
            "visibility": {k: v.numpy().copy() for k, v in b.occ_aware_visibility.items()},
            "events": [(e[0],) + tuple(t.numpy().copy() for t in e[1:]) for e in g.events],
            "n_obs": g.n_obs.numpy().copy(), "collectives": b.shard.collectives, "world": b.shard.world}


def _worker(rank, world, port, args, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = run_backend(*args)
    ret.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_backend_map_static_on_two_ranks_equals_one_process():
    args = (6, [5, 4, 3, 2], [0, 1, 2, 3, 4, 5])          # four window keyframes + two random ones per iteration; densify at iterations 2 and 5, opacity reset at 4
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, args, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = run_backend(*args)
    assert one["world"] == 1 and one["collectives"] == 0
    assert [e[0] for e in one["events"]] == ["densify", "reset", "densify"]
    for rank in (0, 1):
        two = got[rank]
        assert two["world"] == 2 and two["collectives"] > 0
        for a, b in zip(two["params"], one["params"]):                      # the replicas took the same optimizer steps
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)
        for k in one["poses"]:                                               # owner-local pose steps reached every rank
            for a, b in zip(two["poses"][k], one["poses"][k]):
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)
        assert set(two["visibility"]) == set(one["visibility"])
        for k in one["visibility"]:
            assert np.array_equal(two["visibility"][k], one["visibility"][k])
        assert np.array_equal(two["n_obs"], one["n_obs"])
        assert [e[0] for e in two["events"]] == [e[0] for e in one["events"]]
        for e2, e1 in zip(two["events"], one["events"]):                     # reduced statistics (sum / sum / max) and the visibility union
            for a, b in zip(e2[1:], e1[1:]):
                np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=1e-4, atol=1e-7)
    # the two replicas agree with each other exactly: they applied the same reduced gradients
    for a, b in zip(got[0]["params"], got[1]["params"]):
        assert np.array_equal(a, b)
    # a window keyframe is stepped by exactly one rank (uid 0 is never stepped, like in the reference)
    for k in args[1]:
        assert got[0]["steps"][k] + got[1]["steps"][k] == one["steps"][k], k
        assert min(got[0]["steps"][k], got[1]["steps"][k]) == 0
