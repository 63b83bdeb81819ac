"""View-sharded mapping step: one process per GPU, keyframes round-robin over ranks, one
all-reduce(sum) of the flattened per-Gaussian gradient block over RCCL/xGMI before the optimiser step.

The reference has no distributed code (single GPU, two processes); this is the new capability that
BASELINE.json's north_star asks for around the hot path. What is summed is exactly what the reference sums
over the keyframes of one mapping iteration before ``optimizer.step()`` (utils/slam_backend.py:357,526,657,768-771):
the gradients of xyz / features / opacity / scaling / rotation (gaussian_model.py:404-434). Pose gradients stay on
the rank that owns the view (utils/slam_backend.py:955-992).

Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` in the CPU tests. One flat bucket, one collective per step:
xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a single large all-reduce beats many small ones.
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

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
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
        (no pack / unpack copies), through the persistent flat buffer otherwise."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return "single"
        flat = self.grads_as_one_range()
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            return "in-place"
        self.pack()
        self.all_reduce(group)
        self.unpack()
        return "packed"


def allreduce_gaussian_grads(params: Iterable[torch.Tensor], group=None, bucket: GradBucket | None = None) -> GradBucket:
    """pack -> one all-reduce(sum) -> unpack. Re-use the returned bucket across iterations."""
    bucket = bucket or GradBucket(params)
    bucket.all_reduce_grads(group)
    return bucket
