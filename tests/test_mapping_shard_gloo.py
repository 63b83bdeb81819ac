"""world_size-2 gloo test of the view-sharded mapping step: two ranks render different keyframes of the
same Gaussians (oracle-backed stand-in on CPU), all-reduce the flat gradient bucket, and must end with the gradients a
single process gets by summing both views."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import REPO, PKG, make_camera, make_gaussians, make_cotangents, keyframe_pose


def _view_grads(g, k):
    import oracle.torch_binding as ob
    R, t = keyframe_pose(k)
    cam = make_camera(64, 48, R=R, t=t)
    gc, gd = make_cotangents(cam, seed=100 + k)
    T = lambda a, rg=False: torch.tensor(a, requires_grad=rg)
    rs = ob.GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, torch.ones(3), 1.0, T(cam.viewmatrix), T(cam.projmatrix),
                                          T(cam.projmatrix_raw), 0, T(cam.campos), False, False)
    params = [T(g[k_], True) for k_ in ("means3D", "shs", "opacities", "scales", "rotations")]
    P = params[0].shape[0]
    c, r, d, o, n = ob.GaussianRasterizer(rs)(means3D=params[0], means2D=torch.zeros(P, 3, requires_grad=True), opacities=params[2],
                                              shs=params[1], scales=params[3], rotations=params[4])
    ((c * T(gc)).sum() + (d * T(gd)).sum()).backward()
    return params


def _worker(rank, world, port, keyframes, ret):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapping_shard import shard_keyframes, GradBucket
    cam0 = make_camera(64, 48)
    g = make_gaussians(400, cam0, seed=4, scale_mean=0.03)
    mine = shard_keyframes(keyframes, rank, world)
    params = None
    for k in mine:   # accumulate this rank's views
        p = _view_grads(g, k)
        if params is None:
            params = p
        else:
            for a, b in zip(params, p):
                a.grad += b.grad
    bucket = GradBucket(params)
    bucket.pack()
    bucket.all_reduce()
    bucket.unpack()
    if rank == 0:
        ret.put([p.grad.numpy().copy() for p in params])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_sharding_equals_single_process_sum():
    from mapping_shard import shard_keyframes
    keyframes = [0, 1, 2, 3]
    assert shard_keyframes(keyframes, 0, 2) == [0, 2] and shard_keyframes(keyframes, 1, 2) == [1, 3]
    assert sorted(shard_keyframes(list(range(64)), 3, 8)) == list(range(3, 64, 8))
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, keyframes, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cam0 = make_camera(64, 48)
    g = make_gaussians(400, cam0, seed=4, scale_mean=0.03)
    want = None
    for k in keyframes:
        p = _view_grads(g, k)
        want = [x.grad.clone() for x in p] if want is None else [w + x.grad for w, x in zip(want, p)]
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-5, atol=1e-9)


def test_grads_in_one_allocation_are_detected():
    from mapping_shard import GradBucket
    flat = torch.arange(30, dtype=torch.float32)
    ps = [torch.zeros(2, 3, requires_grad=True), torch.zeros(2, 1, 3, requires_grad=True), torch.zeros(2, 1, requires_grad=True)]
    ps[0].grad, ps[1].grad, ps[2].grad = flat[4:10].view(2, 3), flat[10:16].view(2, 1, 3), flat[16:18].view(2, 1)
    r = GradBucket(ps).grads_as_one_range()
    assert r is not None and r.shape == (14,) and r.data_ptr() == flat[4:].data_ptr()
    r += 1                                                   # in place on the gradients' own storage
    assert float(ps[2].grad[1, 0]) == 18.0
    ps[1].grad = torch.zeros(2, 1, 3)                        # not part of the same allocation any more
    assert GradBucket(ps).grads_as_one_range() is None


def test_grad_bucket_single_process_is_identity():
    from mapping_shard import GradBucket, allreduce_gaussian_grads
    ps = [torch.randn(5, 3, requires_grad=True), torch.randn(5, 1, requires_grad=True)]
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    b = allreduce_gaussian_grads(ps)
    assert b.nbytes == 20 * 4
    assert torch.equal(ps[0].grad, torch.full((5, 3), 2.0)) and torch.equal(ps[1].grad, torch.full((5, 1), 3.0))


def _net_worker(rank, world, port, ret):
    """The deformation network's gradients ride the same collective ("plus deformation-net grads")."""
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import deformation
    from mapping_shard import GradBucket
    torch.manual_seed(0)                                          # the same replica on every rank
    net = deformation.deform_network(deformation.default_hidden_params(
        multires=[1, 2], kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [6, 5, 4, 3]}), "cpu")
    params = [p for p in net.parameters() if p.requires_grad]
    gen = torch.Generator().manual_seed(100 + rank)               # different views -> different gradients per rank
    for i, p in enumerate(params):
        if i % 5 != 4:                                            # some parameters see no gradient on a rank (unused heads)
            p.grad = torch.randn(p.shape, generator=gen).contiguous(memory_format=torch.channels_last if p.dim() == 4 else torch.contiguous_format)
    bucket = GradBucket(params)
    mode = bucket.all_reduce_grads()
    if rank == 0:
        ret.put((mode, [None if p.grad is None else p.grad.clone().contiguous().numpy() for p in params], [tuple(p.grad.stride()) if p.grad is not None else None for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_of_deformation_network_gradients():
    import deformation
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29000 + (os.getpid() % 400)
    procs = [ctx.Process(target=_net_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    mode, got, strides = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mode == "packed"                                       # channels-last plane gradients are not one contiguous range
    torch.manual_seed(0)
    net = deformation.deform_network(deformation.default_hidden_params(
        multires=[1, 2], kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [6, 5, 4, 3]}), "cpu")
    params = [p for p in net.parameters() if p.requires_grad]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    for i, p in enumerate(params):
        if i % 5 == 4:
            assert got[i] is not None and float(np.abs(got[i]).max()) == 0.0      # no rank had a gradient: the bucket hands back zeros
            continue
        want = sum(torch.randn(p.shape, generator=g) for g in gens)
        assert np.allclose(got[i], want.numpy(), rtol=1e-6, atol=1e-6), i
        if p.dim() == 4:
            assert strides[i] == tuple(torch.empty(p.shape).contiguous(memory_format=torch.channels_last).stride())   # layout kept


# ---- config #5's step: ShardedMappingStep (attached bucket, no pack / unpack) and the densification-statistics reduction ----------
def _sharded_step_worker(rank, world, port, ret):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapping_shard import ShardedMappingStep, allreduce_densification_stats
    g = torch.Generator().manual_seed(0)                       # identical replicas on every rank
    a = torch.nn.Parameter(torch.randn(50, 3, generator=g))
    planes = torch.nn.Parameter(torch.randn(1, 8, 6, 5, generator=g).contiguous(memory_format=torch.channels_last))   # HexPlane-style layout
    w = torch.nn.Parameter(torch.randn(7, generator=g))
    opt = torch.optim.SGD([a, planes, w], lr=0.1)

    def view_fn(k):        # a "view": a loss that depends on the keyframe id
        ((a * (k + 1)).sum() + (planes * planes).sum() * (k + 2) + (w * k).sum()).backward()

    step = ShardedMappingStep([a, planes, w], list(range(6)), view_fn, optimizer=opt)
    before = [p.detach().clone() for p in (a, planes, w)]
    modes = [step.step() for _ in range(2)]
    # statistics of this rank's views: sum / sum / max over ranks
    acc = torch.full((10, 1), float(rank + 1))
    den = torch.full((10, 1), float(2 * rank + 1))
    rad = torch.arange(10, dtype=torch.float32) * (1 if rank == 0 else -1) + 3 * rank
    allreduce_densification_stats(acc, den, rad)
    if rank == 0:
        ret.put((modes, [p.detach().numpy().copy() for p in (a, planes, w)], [b.numpy() for b in before], planes.grad.stride() == planes.stride(),
                 acc.numpy(), den.numpy(), rad.numpy(), step.keyframes))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mapping_step_and_stats_reduction_two_ranks():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + ((os.getpid() + 137) % 500)
    procs = [ctx.Process(target=_sharded_step_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    modes, after, before, stride_ok, acc, den, rad, kfs = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert modes == ["attached", "attached"] and stride_ok and kfs == [0, 2, 4]
    # single-process reference: all six views, two SGD steps
    a, planes, w = (torch.nn.Parameter(torch.tensor(b)) for b in before)
    opt = torch.optim.SGD([a, planes, w], lr=0.1)
    for _ in range(2):
        opt.zero_grad()
        for k in range(6):
            ((a * (k + 1)).sum() + (planes * planes).sum() * (k + 2) + (w * k).sum()).backward()
        opt.step()
    for got, want in zip(after, (a, planes, w)):
        np.testing.assert_allclose(got, want.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(acc, np.full((10, 1), 3.0))                 # 1 + 2
    np.testing.assert_allclose(den, np.full((10, 1), 4.0))                 # 1 + 3
    np.testing.assert_allclose(rad, np.maximum(np.arange(10.0), -np.arange(10.0) + 3))


# ---- the exchange in two pieces (first n - 1 views reduced asynchronously while the last view runs) equals the single exchange ---------
def _two_piece_worker(rank, world, port, ret):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapping_shard import ShardedMappingStep
    out = {}
    for overlap in (False, True):
        g = torch.Generator().manual_seed(0)                   # identical replicas on every rank, and for both forms
        a = torch.nn.Parameter(torch.randn(40, 3, generator=g))
        planes = torch.nn.Parameter(torch.randn(1, 4, 6, 5, generator=g).contiguous(memory_format=torch.channels_last))
        opt = torch.optim.Adam([a, planes], lr=0.05)

        def view_fn(k, a=a, planes=planes):
            (torch.sin(a * (k + 1)).sum() + (planes ** 3).sum() * (0.1 * k + 0.2)).backward()

        step = ShardedMappingStep([a, planes], list(range(8)), view_fn, optimizer=opt, overlap=overlap)
        modes = [step.step() for _ in range(3)]
        out[overlap] = (modes, a.detach().numpy().copy(), planes.detach().numpy().copy(), step.allreduce_calls, planes.grad.stride() == planes.stride())
    if rank == 0:
        ret.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_piece_exchange_equals_single_exchange():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29300 + (os.getpid() % 150)
    procs = [ctx.Process(target=_two_piece_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    out = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one, two = out[False], out[True]
    assert one[0] == ["attached"] * 3 and two[0] == ["two-piece"] * 3
    assert one[3] == 3 and two[3] == 6                       # one collective per step vs two (the first of them asynchronous)
    assert two[4]                                            # the channels-last gradient kept the parameter's strides
    np.testing.assert_allclose(two[1], one[1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(two[2], one[2], rtol=1e-5, atol=1e-6)


# ---- SURVEY.md 8e's alternative exchange: reduce-scatter -> Adam on the local 1/N slice -> all-gather of the parameters ----------------------
class _FlatAdam:
    """Test double of FusedAdam's two entry points on CPU tensors: plain Adam arithmetic on a range of the flat parameter / gradient
    buffers (step_slice), or on everything (step). Moments live in flat buffers of its own."""

    def __init__(self, params, lr=0.05, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr, self.betas, self.eps, self.t = list(params), lr, betas, eps, 0
        n = sum(p.numel() for p in self.params)
        self.m, self.v = torch.zeros(n + 8), torch.zeros(n + 8)

    def _apply(self, p_flat, g_flat, lo, hi):
        b1, b2 = self.betas
        m, v = self.m[lo:hi], self.v[lo:hi]
        g = g_flat[lo:hi]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p_flat[lo:hi] -= (self.lr / (1 - b1 ** self.t)) * m / (v.sqrt() / (1 - b2 ** self.t) ** 0.5 + self.eps)

    @torch.no_grad()
    def step_slice(self, param_bucket, lo, hi):
        self.t += 1
        g_flat = torch.cat([p.grad.reshape(-1) for p in param_bucket.params])
        g_flat = torch.cat([g_flat, torch.zeros(param_bucket.flat.numel() - g_flat.numel())])
        self._apply(param_bucket.flat, g_flat, lo, hi)

    @torch.no_grad()
    def step(self):
        self.t += 1
        o = 0
        for p in self.params:
            n = p.numel()
            flat = p.data.view(-1)
            b1, b2 = self.betas
            m, v, g = self.m[o:o + n], self.v[o:o + n], p.grad.reshape(-1)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            flat -= (self.lr / (1 - b1 ** self.t)) * m / (v.sqrt() / (1 - b2 ** self.t) ** 0.5 + self.eps)
            o += n


def _exchange_worker(rank, world, port, ret):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapping_shard import ShardedMappingStep
    out = {}
    for exchange in ("all_reduce", "reduce_scatter"):
        g = torch.Generator().manual_seed(0)                   # identical replicas on every rank, and for both exchanges
        a = torch.nn.Parameter(torch.randn(41, 3, generator=g))
        b = torch.nn.Parameter(torch.randn(7, generator=g))
        c = torch.nn.Parameter(torch.randn(7, 3, generator=g))            # 151 elements in all: padded to 152 for two equal pieces
        opt = _FlatAdam([a, b, c])
        seen = []

        def views_fn(ks, a=a, b=b, c=c, seen=seen):             # the multi-view form: this rank's keyframes in ONE call
            seen.append(list(ks))
            for k in ks:
                (torch.sin(a * (k + 1)).sum() + (b ** 3).sum() * (0.1 * k + 0.2) + torch.cos(c * 0.5 * (k + 1)).sum()).backward()

        step = ShardedMappingStep([a, b, c], list(range(8)), None, optimizer=opt, exchange=exchange, views_fn=views_fn)
        modes = [step.step() for _ in range(3)]
        out[exchange] = (modes, [t.detach().numpy().copy() for t in (a, b, c)], step.allreduce_calls, seen[0],
                         getattr(step, "slice", None), int(step.bucket.flat.numel()))
    ret.put((rank, out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_reduce_scatter_exchange_equals_all_reduce_exchange():
    """reduce-scatter -> optimizer on this rank's 1/N slice of the flat parameter buffer -> all-gather of the parameters
    must leave every rank with the parameters the all-reduce exchange leaves, after 3 steps, and both must equal one process."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29100 + (os.getpid() % 150)
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    solo = ctx.Queue()
    _exchange_worker(0, 1, port + 1, solo)
    one = solo.get(timeout=10)[1]
    assert one["all_reduce"][0] == ["single"] * 3 and one["reduce_scatter"][0] == ["single"] * 3
    for rank in (0, 1):
        ar, rs = got[rank]["all_reduce"], got[rank]["reduce_scatter"]
        assert ar[0] == ["attached"] * 3 and rs[0] == ["reduce-scatter"] * 3
        assert rs[3] == [rank, rank + 2, rank + 4, rank + 6]              # this rank's keyframes arrived as one list
        assert rs[5] == 152 and rs[4] == (76 * rank, 76 * (rank + 1))      # padded bucket, equal pieces
        for x, y, z in zip(ar[1], rs[1], one["all_reduce"][1]):
            np.testing.assert_allclose(y, x, rtol=1e-6, atol=1e-7)         # the two exchanges agree ...
            np.testing.assert_allclose(x, z, rtol=1e-5, atol=1e-6)         # ... and equal the single process
    for x, y in zip(got[0]["reduce_scatter"][1], got[1]["reduce_scatter"][1]):
        assert np.array_equal(x, y)                                        # the replicas hold the same parameters after the all-gather


# ---- ViewShard.reduce_gradients: packed path (cached bucket, "None stays None") and the attached network bucket; bit-packed visibility rows ----
def _viewshard_worker(rank, world, port, ret):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapping_shard import ViewShard
    shard = ViewShard()
    g = torch.Generator().manual_seed(1)
    a, b, c = (torch.nn.Parameter(torch.randn(5, 3, generator=g)), torch.nn.Parameter(torch.randn(4, generator=g)), torch.nn.Parameter(torch.randn(2, 2, generator=g)))
    opt = torch.optim.Adam([a, b, c], lr=0.1)
    # packed path: c has no gradient on either rank, b only on rank 1
    ((a * (rank + 1)).sum()).backward()
    if rank == 1:
        (b * 3.0).sum().backward()
    shard.reduce_gradients(opt)
    bucket_1 = shard.__dict__["_gauss_pack"][1]
    packed = (a.grad.clone(), None if b.grad is None else b.grad.clone(), c.grad is None)
    opt.step()
    state_has_c = c in opt.state and len(opt.state[c]) > 0
    opt.zero_grad(set_to_none=True)
    ((a * 2.0).sum()).backward()
    shard.reduce_gradients(opt)
    same_bucket = shard.__dict__["_gauss_pack"][1] is bucket_1                      # cached, not rebuilt per iteration
    # attached network bucket: gradients accumulate in place, reduced without packing
    # (the third parameter never receives a gradient -- the detached `nodes` group of the deform model: it must end with grad None and no
    # optimizer state, as in a single process. Liveness is read off `grad is None` on a first PACKED iteration, never off
    # values: the fourth parameter is in the graph with an all-zero gradient -- a single process steps it, so it must be attached)
    net = [torch.nn.Parameter(torch.ones(3, 2)), torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(2))]
    loss = lambda: (net[0] * (rank + 1)).sum() + (net[1] * 10.0 * (rank + 1)).sum() + (net[3] * 0.0).sum()
    probe = shard.attach_network(net) is None and all(p.grad is None for p in net)      # unseen parameter list: the first iteration is the probe
    loss().backward()
    before = shard.collectives
    shard.reduce_gradients(None, net)
    net_opt = torch.optim.Adam(net, lr=0.1)
    net_opt.step()
    probe = probe and shard.collectives - before == 1 and net[2].grad is None and float(net[0].grad.sum()) == 18.0 and float(net[3].grad.abs().sum()) == 0.0
    shard.zero_network_grads(net_opt)                                                  # ... and attaches the live parameters
    nb = shard._net_bucket[1]
    views = [p.grad for p in net]
    loss().backward()
    still_views = all(p.grad is v for p, v in zip(net, views)) and views[2] is None and views[3] is not None
    before = shard.collectives
    shard.reduce_gradients(None, net)
    net_opt.step()
    untouched_ok = (probe and net[2].grad is None and [id(p) for p in nb.params] == [id(net[0]), id(net[1]), id(net[3])] and len(net_opt.state.get(net[2], {})) == 0
                    and float(net[2].detach().sum()) == 5.0 and int(net_opt.state[net[3]]["step"]) == 2)
    attached = (net[0].grad.clone(), net[1].grad.clone(), shard.collectives - before, still_views and untouched_ok)
    views = [p.grad for p in (net[0], net[1], net[3])]
    shard.zero_network_grads(net_opt)
    zeroed = float(nb.flat.abs().sum()) == 0.0 and all(p.grad is v for p, v in zip((net[0], net[1], net[3]), views)) and net[2].grad is None
    # next iteration: still the attached path (one collective), the left-out parameter still None
    loss().backward()
    before = shard.collectives
    shard.reduce_gradients(None, net)
    zeroed = zeroed and shard.collectives - before == 1 and net[2].grad is None and float(net[0].grad.sum()) == 18.0
    # ... and if it does receive a gradient later, that iteration goes through the packed path, nothing is lost, and the next
    # zero_network_grads() rebuilds the bucket WITH it (no stale gradient survives into the iteration after)
    shard.zero_network_grads(net_opt)
    (net[2] * (rank + 1)).sum().backward()
    shard.reduce_gradients(None, net)
    zeroed = zeroed and net[2].grad is not None and float(net[2].grad.sum()) == 15.0 and id(net[2]) not in shard._net_no_grad
    shard.zero_network_grads(net_opt)
    nb2 = shard._net_bucket[1]
    zeroed = zeroed and id(net[2]) in [id(p) for p in nb2.params] and all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in (net[2], net[3]))
    # bit-packed 0 / 1 rows: 3 rows of 21 flags, owner = index % world
    rows_all = [(torch.arange(21) % (k + 2) == 0).long() for k in range(3)]
    got = shard.gather_mask_rows({k: rows_all[k] for k in range(3) if shard.owns(k)}, 3, 21, torch.device("cpu"))
    rows_ok = all(torch.equal(r, w) for r, w in zip(got, rows_all))
    N = lambda t: None if t is None else (t.numpy().copy() if torch.is_tensor(t) else t)          # (tensors do not survive the queue once the worker exits)
    ret.put((rank, tuple(N(t) for t in packed), state_has_c, same_bucket, tuple(N(t) for t in attached), zeroed, rows_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_viewshard_packed_and_attached_network_paths_and_bit_packed_rows():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29250 + (os.getpid() % 40)
    procs = [ctx.Process(target=_viewshard_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r[0], r[1:]) for r in (ret.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        packed, state_has_c, same_bucket, attached, zeroed, rows_ok = got[rank]
        assert np.array_equal(packed[0], np.full((5, 3), 3.0, np.float32))           # 1 + 2
        assert packed[1] is not None and np.array_equal(packed[1], np.full((4,), 3.0, np.float32))   # rank 1's gradient reached rank 0
        assert packed[2] and not state_has_c                                         # no gradient anywhere: stays None, no Adam state created
        assert same_bucket
        assert np.array_equal(attached[0], np.full((3, 2), 3.0, np.float32)) and np.array_equal(attached[1], np.full((4,), 30.0, np.float32))
        assert attached[2] == 1 and attached[3]                                      # ONE collective, on the gradients' own storage
        assert zeroed and rows_ok
