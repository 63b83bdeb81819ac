"""View-sharded mapping step: one process per GPU, keyframes round-robin over ranks, one
all-reduce(sum) of the flattened per-Gaussian gradient block over RCCL/xGMI before the optimiser step.

The reference has no distributed code (single GPU, two processes); this is the new capability that
BASELINE.json's north_star asks for around the hot path. What is summed is exactly what the reference sums
over the keyframes of one mapping iteration before ``optimizer.step()`` (utils/slam_backend.py:357,526,657,768-771):
the gradients of xyz / features / opacity / scaling / rotation (gaussian_model.py:404-434). Pose gradients stay on
the rank that owns the view (utils/slam_backend.py:955-992).

Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` in the CPU tests. One flat bucket, one collective per step:
xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a single large all-reduce beats many small ones.

What can overlap with that all-reduce: the sum over views is linear, every parameter's gradient is complete only when the LAST view's
geometry kernel has run (the fused K8+K9 kernel writes all five tensors at once), and the optimizer needs the reduced sum before the
next forward pass -- so a full-size exchange always follows the last backward pass. ``ShardedMappingStep(overlap=True)`` nevertheless
splits the exchange in two (the first n - 1 views' sum goes out asynchronously while the last view renders into a second bucket):
it hides the first collective's start-up and rank imbalance, not bytes; it is off by default and ``bench.py`` reports both forms when it
runs on more than one GPU. `GradBucket.attach()` removes the other cost -- pack / unpack copies -- for every parameter layout,
channels-last included.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_keyframes(keyframe_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """Round-robin ownership: keyframe k belongs to rank k % world_size (cfg #5: 64 keyframes -> 8 per GPU)."""
    return [k for i, k in enumerate(keyframe_ids) if i % world_size == rank]


class GradBucket:
    """Persistent flat fp32 buffer holding the gradients of a fixed parameter list, all-reduced in one call."""

    def __init__(self, params: Iterable[torch.Tensor], pad_to: int = 1):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        self.used = total
        total = (total + pad_to - 1) // pad_to * pad_to          # (reduce-scatter needs equal pieces: zero padding behind the last tensor)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[o:o + n].view_as(p))
            o += n

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    # ---- persistent gradient storage: the parameters' .grad tensors ARE views of the flat buffer ---------------------------------
    def attach(self, fused_accumulate: bool = True):
        """Make every parameter's ``.grad`` a view of the flat buffer with the PARAMETER's own strides (so channels-last HexPlane planes,
        whose dense gradient is not one contiguous row-major range, qualify too). autograd accumulates into an existing ``.grad`` in
        place, so after any number of backward passes the flat buffer holds all gradients back to back and ``all_reduce()`` needs no
        pack / unpack copy. Call ``zero_grads()`` instead of ``optimizer.zero_grad(set_to_none=True)`` between iterations.
        fused_accumulate: additionally mark the parameters so that the rasterizer's backward kernels add each view's gradients to these
        buffers themselves (diff_gaussian_rasterization.autograd._accumulation_targets) -- no AccumulateGrad pass over five tensors per
        view, no zero rows for invisible Gaussians. Only for loss.backward() style training; torch.autograd.grad() needs it off."""
        self.views = []
        o = 0
        for p, n in zip(self.params, self.sizes):
            dense = p.numel() == 0 or _is_dense(p)
            if not dense:
                raise ValueError("GradBucket.attach: parameter with overlapping / gapped memory")
            v = torch.as_strided(self.flat, p.shape, p.stride(), o) if p.numel() else self.flat[o:o].view(p.shape)
            self.views.append(v)
            p.grad = v
            setattr(p, "_gsr_accumulate_grad", bool(fused_accumulate))
            o += n
        self.attached = True
        return self

    def zero_grads(self):
        self.flat.zero_()
        if getattr(self, "attached", False):
            for p, v in zip(self.params, self.views):
                if p.grad is not v:            # something (an optimizer's zero_grad(set_to_none=True)) dropped the view: put it back
                    p.grad = v

    def pack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def unpack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def all_reduce(self, group=None, async_op: bool = False):
        """Sum over ranks. Returns the work handle when async_op (so the collective can overlap the next view)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    def grads_as_one_range(self):
        """If the parameters' .grad tensors already sit back to back in one allocation, in parameter order (the rasterizer's
        backward returns means3D / sh / opacity / scales / rotations gradients that way), return a flat view over them, else None."""
        gs = [p.grad for p in self.params]
        if not gs or any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in gs):
            return None
        st = gs[0].untyped_storage().data_ptr()
        off = gs[0].storage_offset()
        for g in gs:
            if g.untyped_storage().data_ptr() != st or g.storage_offset() != off:
                return None
            off += g.numel()
        return gs[0].as_strided((off - gs[0].storage_offset(),), (1,), gs[0].storage_offset())

    def all_reduce_grads(self, group=None):
        """Sum the parameters' gradients over ranks: in place on the gradients' own storage when they form one range
        (no pack / unpack copies), through the persistent flat buffer otherwise. A one-rank group skips the collective unless
        GSR_FORCE_COLLECTIVE=1 (tests: the RCCL call on the real buffers, on a box with a single GPU)."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _force_collective()):
            return "single"
        if getattr(self, "attached", False) and all(p.grad is v for p, v in zip(self.params, self.views)):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            return "attached"
        flat = self.grads_as_one_range()
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            return "in-place"
        self.pack()
        self.all_reduce(group)
        self.unpack()
        return "packed"


class ParamBucket:
    """The PARAMETERS of a fixed list back to back in one flat fp32 buffer (each parameter's .data becomes a view of it, same values), laid
    out like GradBucket lays out their gradients: what an all-gather of updated parameter slices needs (reduce-scatter exchange below)."""

    def __init__(self, params: Iterable[torch.Tensor], pad_to: int = 1):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        total = (sum(self.sizes) + pad_to - 1) // pad_to * pad_to
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        o = 0
        with torch.no_grad():
            for p, n in zip(self.params, self.sizes):
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise ValueError("ParamBucket: contiguous float32 parameters only")
                v = self.flat[o:o + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.offsets.append(o)
                o += n


def _force_collective() -> bool:
    import os
    return os.environ.get("GSR_FORCE_COLLECTIVE") == "1"


def allreduce_gaussian_grads(params: Iterable[torch.Tensor], group=None, bucket: GradBucket | None = None) -> GradBucket:
    """pack -> one all-reduce(sum) -> unpack. Re-use the returned bucket across iterations."""
    bucket = bucket or GradBucket(params)
    bucket.all_reduce_grads(group)
    return bucket


def _is_dense(t: torch.Tensor) -> bool:
    """True if t's elements occupy exactly numel() consecutive storage slots in some dimension order (contiguous, channels-last, ...)."""
    dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1), key=lambda x: x[0])
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


def allreduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor, group=None):
    """The cross-rank reduction of the densification statistics: every rank accumulated them over ITS views
    (gaussian_model.py:973-977, utils/slam_backend.py:714-721), so before densify_and_prune
        xyz_gradient_accum, denom -> sum over ranks (one collective over both, packed),   max_radii2D -> max over ranks.
    In place; call it only on the iterations that densify (every ``gaussian_update_every``), not every step."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _force_collective()):
        return
    P = xyz_gradient_accum.numel()
    packed = torch.cat([xyz_gradient_accum.reshape(-1), denom.reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    xyz_gradient_accum.copy_(packed[:P].view_as(xyz_gradient_accum))
    denom.copy_(packed[P:].view_as(denom))
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


class ViewShard:
    """Who renders which view of a mapping iteration, and the collectives that put the pieces back together.

    Every rank holds a replica of the map and runs the SAME host program (same window, same random draws: seed every rank alike); a view
    with index k in the iteration's view list is rendered, back-propagated and pose-stepped by rank ``k % world`` only. Per iteration:

        reduce_gradients()    ONE all-reduce(sum) of the Gaussian gradients (the optimizer's flat bucket when FusedAdam keeps one) and one
                              of the deformation / node network's, before the optimizer steps -- every replica then applies the same update;
        sync_cameras()        the window cameras' poses and exposures, owner -> everyone (14 floats per camera, one small all-reduce);
    and only on the iterations that need them:
        reduce_statistics()   densification statistics: sum / sum / max over ranks, before densify_and_prune;
        union()               a visibility mask OR-ed over ranks (opacity reset of the Gaussians no view saw);
        gather_rows()         per-view rows (n_touched > 0 of the window keyframes) from their owners to everyone.
    With one rank (or no process group) every method is a no-op that returns its input. (Round 3 also had replicate_gradients(): loops that
    every rank runs redundantly stepped on rank 0's gradients, because torch's scatter backward passes were not reproducible. The dynamic
    branch is bit-reproducible now -- control_nodes.gather_rows, the ordered node-blend backward -- and the broadcast is gone.)"""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self._net_bucket = None
        self.collectives = 0

    @property
    def active(self) -> bool:
        return self.world > 1

    def owns(self, index: int) -> bool:
        return index % self.world == self.rank

    def owner(self, index: int) -> int:
        return index % self.world

    def _sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.collectives += 1
        return t

    def reduce_gradients(self, optimizer, network_params=()):
        """Gaussian gradients through the optimizer's flat bucket when it has one (FusedAdam with fused accumulation: the kernels already
        summed this rank's views into it), else through a cached packing bucket; the network's gradients ride a second bucket (different
        lifetime: the Gaussian bucket is rebuilt by every densification) -- in place when attach_network() made the gradients views of it,
        packed otherwise. On the PACKED path a parameter that has no gradient on ANY rank keeps ``grad = None`` (no optimizer state is created
        for it, as in a single process); attach_network() attaches only parameters that already received a gradient for the same reason.
        Every rank must take the same path for the same parameters (a precondition: see _packed_sum)."""
        if not self.active:
            return
        bucket = getattr(optimizer, "_bucket", None)
        params = [p for g_ in optimizer.param_groups for p in g_["params"]] if optimizer is not None else []
        if optimizer is None:
            pass
        elif bucket is not None and getattr(bucket, "attached", False) and all(p.grad is v for p, v in zip(bucket.params, bucket.views)) \
                and len(bucket.params) == len(params):
            self._sum(bucket.flat)
        else:
            self._packed_sum("_gauss_pack", params)
        net = [p for p in network_params if p.requires_grad]
        if net:
            hit = self._net_bucket
            key = tuple(id(p) for p in net)
            dead = self.__dict__.setdefault("_net_no_grad", set())
            if hit is not None and hit[0] == key and getattr(hit[1], "attached", False) \
                    and all(p.grad is v for p, v in zip(hit[1].params, hit[1].views)) and all(p.grad is None for p in net if id(p) in dead):
                self._sum(hit[1].flat)
            else:
                # Packed iteration: also the PROBE that decides which parameters the attached bucket will hold. A parameter whose .grad is
                # None on every rank (it stays None through _packed_sum) takes no part in the loss -- the detached `nodes` group, heads the
                # configuration switches off; Adam skips it in a single process and must here, so it stays out of the bucket (an attached
                # view is zero rather than None). Liveness comes from None-ness alone: a parameter that is in the graph with an all-zero
                # gradient holds a zero tensor, is stepped by a single process, and is attached here (value == 0 was the old test).
                self._packed_sum("_net_pack", net)
                dead.clear()
                dead.update(id(p) for p in net if p.grad is None)
                self._net_probed = key
                if hit is not None and getattr(hit[1], "attached", False) and [id(p) for p in hit[1].params] != [id(p) for p in net if id(p) not in dead]:
                    hit[1].attached = False              # its parameter list is out of date: zero_network_grads() rebuilds and re-attaches

    def _packed_sum(self, slot, params):
        """pack -> ONE all-reduce -> unpack through a bucket cached per parameter list; gradients that are None on every rank stay None.
        Which parameters hold a gradient anywhere travels IN the same collective (one flag per parameter behind the packed gradients), and the
        host reads the flags back only if THIS rank has a parameter without a gradient -- the usual iteration (every rank runs the same graph,
        every parameter has one) costs no extra collective and no host synchronisation. Precondition (documented, not checkable from inside:
        mismatched collectives hang or fail in the backend rather than raise here): every rank takes the same path -- attached bucket or packed
        -- for the same parameter list in the same iteration."""
        key = tuple(id(p) for p in params)
        hit = self.__dict__.get(slot)
        if hit is None or hit[0] != key:
            b = GradBucket(params)
            hit = (key, b, torch.zeros(b.flat.numel() + len(params), dtype=torch.float32, device=b.flat.device))
            self.__dict__[slot] = hit
        b, wire = hit[1], hit[2]
        n = b.flat.numel()
        b.pack()
        wire[:n].copy_(b.flat)
        missing = [k for k, p in enumerate(params) if p.grad is None]
        wire[n:].fill_(1.0)
        if missing:
            wire[n:][torch.tensor(missing, device=wire.device)] = 0.0
        self._sum(wire)
        b.flat.copy_(wire[:n])
        holders = wire[n:].tolist() if missing else None           # (host read only when some local gradient is None)
        for k, (p, v) in enumerate(zip(b.params, b.views)):
            if holders is not None and holders[k] == 0:
                p.grad = None
            elif p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def attach_network(self, params):
        """Make the network parameters' .grad views of one persistent flat bucket (GradBucket.attach without the rasterizer's fused
        accumulation): autograd accumulates into them in place and reduce_gradients all-reduces the bucket as it stands -- no pack / unpack
        of ~20 tensors per iteration. Use zero_network_grads() in place of optimizer.zero_grad(set_to_none=True). No-op with one rank.
        Only parameters that RECEIVE gradients are attached: an attached view is zero rather than None, and Adam would step a parameter the
        loss never reaches with g = 0 and create state for it. Which ones those are is read off ``p.grad is None`` after one PACKED
        iteration: the first iteration of a parameter list this shard has not seen runs unattached (returns None), reduce_gradients probes,
        and the first zero_network_grads() behind it attaches the live parameters -- same optimizer state as a single process."""
        every = [p for p in params if p.requires_grad]
        if not self.active or not every:
            return None
        key = tuple(id(p) for p in every)
        self._net_want = (key, every)
        if self.__dict__.get("_net_probed") != key:
            return None
        return self._attach_live()

    def _attach_live(self):
        key, every = self._net_want
        dead = self.__dict__.setdefault("_net_no_grad", set())
        net = [p for p in every if id(p) not in dead]
        for p in every:
            if id(p) in dead:
                p.grad = None
        if not net:
            return None
        if self._net_bucket is None or self._net_bucket[0] != key or [id(p) for p in self._net_bucket[1].params] != [id(p) for p in net]:
            self._net_bucket = (key, GradBucket(net))
        b = self._net_bucket[1]
        b.flat.zero_()
        b.attach(fused_accumulate=False)
        return b

    def zero_network_grads(self, optimizer):
        hit, want = self._net_bucket, self.__dict__.get("_net_want")
        if self.active and want is not None and self.__dict__.get("_net_probed") == want[0]:
            dead = self.__dict__.setdefault("_net_no_grad", set())
            live = [id(p) for p in want[1] if id(p) not in dead]
            if hit is not None and hit[0] == want[0] and getattr(hit[1], "attached", False) and [id(p) for p in hit[1].params] == live:
                hit[1].zero_grads()
            else:                                        # behind the probing iteration, or a parameter joined / left: (re)build and attach
                for p in want[1]:
                    p.grad = None
                    if hasattr(p, "_gsr_accumulate_grad"):
                        delattr(p, "_gsr_accumulate_grad")
                self._attach_live()
        elif self.active and hit is not None and getattr(hit[1], "attached", False):
            hit[1].zero_grads()
        else:
            optimizer.zero_grad(set_to_none=True)

    def reduce_statistics(self, gaussians):
        if self.active:
            allreduce_densification_stats(gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D, self.group)
            self.collectives += 2

    def union(self, mask):
        if not self.active:
            return mask
        m = mask.to(torch.int32)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        self.collectives += 1
        return m.to(mask.dtype)

    def gather_rows(self, rows, count, like):
        """rows: {index: 1-D tensor} for the indices this rank owns, each shaped and typed like `like`; returns the list of all `count` rows."""
        if not self.active:
            return [rows[k] for k in range(count)]
        full = torch.zeros((count,) + tuple(like.shape), dtype=like.dtype, device=like.device)
        for k, r in rows.items():
            full[k] = r
        self._sum(full)
        return [full[k] for k in range(count)]

    def gather_mask_rows(self, rows, count, length, device):
        """gather_rows for 0 / 1 rows (the window keyframes' visibility, 1 bit per Gaussian and keyframe): every rank packs the
        rows it owns 8 to a byte, the packed [count, length / 8] matrix is summed over ranks (each row has ONE owner, the others contribute
        zeros) and unpacked -- 64x fewer bytes than the int64 rows. Returns `count` int64 rows of 0 / 1."""
        if not self.active:
            return [(rows[k] != 0).long() for k in range(count)]
        nbytes = (length + 7) // 8
        weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=device)
        packed = torch.zeros((count, nbytes), dtype=torch.int32, device=device)
        for k, r in rows.items():
            bits = torch.zeros(nbytes * 8, dtype=torch.int32, device=device)
            bits[:length] = (r != 0).to(torch.int32)
            packed[k] = (bits.view(nbytes, 8) * weights).sum(dim=1)
        packed = packed.to(torch.uint8)
        self._sum(packed)
        bits = (packed.to(torch.int32)[:, :, None] // weights) % 2                     # [count, nbytes, 8]
        full = bits.reshape(count, nbytes * 8)[:, :length].long()
        return [full[k] for k in range(count)]

    def sync_cameras(self, cameras):
        """cameras: the iteration's view list (index = ownership index). Owner -> everyone: R, T, exposure_a, exposure_b."""
        if not self.active or not cameras:
            return
        dev = cameras[0].R.device
        pack = torch.zeros((len(cameras), 14), dtype=torch.float32, device=dev)
        for k, c in enumerate(cameras):
            if self.owns(k):
                pack[k, :9] = c.R.reshape(-1)
                pack[k, 9:12] = c.T.reshape(-1)
                pack[k, 12] = c.exposure_a.detach().reshape(-1)[0]
                pack[k, 13] = c.exposure_b.detach().reshape(-1)[0]
        self._sum(pack)
        with torch.no_grad():
            for k, c in enumerate(cameras):
                if not self.owns(k):
                    c.update_RT(pack[k, :9].view(3, 3).clone(), pack[k, 9:12].clone())
                    c.exposure_a.copy_(pack[k, 12].view_as(c.exposure_a))
                    c.exposure_b.copy_(pack[k, 13].view_as(c.exposure_b))


class ShardedMappingStep:
    """One mapping iteration over a set of keyframes, view-sharded (BASELINE config #5):

        step():  zero the gradient bucket
                 for k in this rank's keyframes:  view_fn(k)      # render + loss + backward; gradients accumulate in the bucket
                 ONE all-reduce(sum) of the bucket                # 14 floats per Gaussian at SH degree 0: 112 MB at 2 M Gaussians
                 optimizer.step()                                 # every rank applies the same update to its replica

    ``params`` in the optimizer's order; ``view_fn(k)`` must leave pose / exposure gradients alone (they belong to the owner)."""

    def __init__(self, params, keyframe_ids, view_fn, optimizer=None, group=None, overlap=False, exchange="all_reduce", views_fn=None, local=False):
        """exchange = "all_reduce" (default): every rank receives the full gradient sum and steps its whole replica.
        exchange = "reduce_scatter" (SURVEY.md 8e's alternative): reduce-scatter of the gradient bucket -> the optimizer steps only this
        rank's 1/world slice of the flat parameter buffer (FusedAdam.step_slice) -> all-gather of the updated parameter slices. The same
        2 (N - 1) / N bucket sizes cross every link, but only the reduce-scatter half stands between the last backward pass and the
        optimizer, Adam's traffic is divided by N, and on a full xGMI mesh both halves are direct exchanges (every rank sends piece r to rank r)
        instead of a ring. Needs an optimizer with step_slice (FusedAdam; the CPU tests pass a plain-torch stand-in). Moments outside a
        rank's slice are never touched on that rank.
        views_fn: optional callable taking the LIST of this rank's keyframes (the multi-view entry point renders and back-propagates them
        with one launch per pipeline stage); when absent, view_fn is called per keyframe.
        local: ignore the process group -- every keyframe on this rank, no collective (the single-GPU point of a scaling curve, run by one
        rank of a larger job)."""
        self.group = group
        self.local = bool(local)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized() and not local) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.keyframes = shard_keyframes(list(keyframe_ids), self.rank, self.world)
        self.view_fn = view_fn
        self.views_fn = views_fn
        self.optimizer = optimizer
        if exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError("ShardedMappingStep: exchange must be 'all_reduce' or 'reduce_scatter'")
        self.exchange = exchange
        params = list(params)
        if exchange == "reduce_scatter":
            if overlap:
                raise ValueError("ShardedMappingStep: the two-piece overlap belongs to the all-reduce exchange")
            if optimizer is None or not hasattr(optimizer, "step_slice"):
                raise ValueError("ShardedMappingStep: exchange='reduce_scatter' needs an optimizer with step_slice(param_bucket, lo, hi)")
            self.param_bucket = ParamBucket(params, pad_to=self.world)
            self.bucket = GradBucket(params, pad_to=self.world).attach()
            piece = self.bucket.flat.numel() // self.world
            self.slice = (self.rank * piece, (self.rank + 1) * piece)
        else:
            self.bucket = GradBucket(params).attach()
        # overlap (SURVEY.md 8e "Overlap"): the exchange in two pieces -- the sum of this rank's first n - 1 views is all-reduced
        # asynchronously WHILE the last view renders into a second bucket, which is reduced behind it; the two reduced pieces are added.
        # What it buys is bounded: the second piece is as large as the whole bucket, so the bytes behind the last backward pass are the
        # same; only the latency of the first collective's start-up and any imbalance between ranks hide behind the last view.
        self.overlap = bool(overlap) and len(self.keyframes) >= 2
        self.last_bucket = GradBucket(params) if self.overlap else None
        self.mode = None
        self.allreduce_calls = 0

    def _collectives_on(self):
        return not self.local and dist.is_available() and dist.is_initialized() and (self.world > 1 or _force_collective())

    def _render_mine(self, keyframes):
        if self.views_fn is not None:
            if keyframes:
                self.views_fn(list(keyframes))
        else:
            for k in keyframes:
                self.view_fn(k)

    def step(self):
        if self.exchange == "reduce_scatter":
            self.bucket.zero_grads()
            self._render_mine(self.keyframes)
            lo, hi = self.slice
            if self._collectives_on():
                dist.reduce_scatter_tensor(self.bucket.flat[lo:hi], self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.allreduce_calls += 1
            self.optimizer.step_slice(self.param_bucket, lo, hi)
            if self._collectives_on():
                dist.all_gather_into_tensor(self.param_bucket.flat, self.param_bucket.flat[lo:hi], group=self.group)
            self.mode = "reduce-scatter" if self._collectives_on() else "single"
            return self.mode
        if not self.overlap:
            self.bucket.zero_grads()
            self._render_mine(self.keyframes)
            self.mode = "single" if self.local else self.bucket.all_reduce_grads(self.group)
            self.allreduce_calls += 0 if self.mode == "single" else 1
        else:
            first, last = self.bucket, self.last_bucket
            first.attach()
            first.flat.zero_()
            self._render_mine(self.keyframes[:-1])
            work = dist.all_reduce(first.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self._collectives_on() else None
            last.attach()                                    # the parameters' .grad now point into the second bucket
            last.flat.zero_()
            self._render_mine(self.keyframes[-1:])
            if self._collectives_on():
                dist.all_reduce(last.flat, op=dist.ReduceOp.SUM, group=self.group)
                work.wait()
                self.allreduce_calls += 2
            first.flat.add_(last.flat)
            first.attach()                                   # the optimizer reads the total through the parameters' .grad
            self.mode = "two-piece"
        if self.optimizer is not None:
            self.optimizer.step()
        return self.mode
