"""Fused Adam for the Gaussian model (second half): a ``torch.optim.Adam`` whose ``step()`` updates every
parameter tensor in ONE HIP launch (include/slam_losses.h, gsr_adam_step). It subclasses torch.optim.Adam and keeps its state
layout (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``, ``param_groups`` with per-group ``lr`` / ``name``), so the reference's
densification code -- ``replace_tensor_to_optimizer``, ``cat_tensors_to_optimizer``, ``_prune_optimizer``
(scene/gaussian_model.py:734-865), which edit ``optimizer.state`` and ``param_groups`` directly -- and its learning-rate schedule
(:492-505) work unchanged:

    self.optimizer = FusedAdam(l, lr=0.0, eps=1e-15)        # instead of torch.optim.Adam(l, lr=0.0, eps=1e-15), :447

Only what the reference uses is fused (no amsgrad, weight decay, maximize, capturable); anything else, more than 32 tensors with
gradients, or non-float32 / non-contiguous tensors goes through torch.optim.Adam's own step()."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C

_MAX_SEGMENTS = 32


class _Segment(C.Structure):        # gsr_adam_segment
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_ulonglong), ("lr", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("beta1_d", C.c_double),
                ("beta2_d", C.c_double), ("step", C.c_int)]


_declared = False


def _lib():
    global _declared
    lib = _C.load_library()
    if not _declared:
        lib.gsr_adam_step.restype = C.c_int
        lib.gsr_adam_step.argtypes = [C.c_int, C.POINTER(_Segment), C.c_void_p]
        lib.gsr_adam_step_scheduled.restype = C.c_int
        lib.gsr_adam_step_scheduled.argtypes = [C.c_int, C.POINTER(_Segment), C.c_void_p, C.c_void_p]
        lib.gsr_adam_step_device_count.restype = C.c_int
        lib.gsr_adam_step_device_count.argtypes = [C.c_int, C.POINTER(_Segment), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
        lib.gsr_adam_coefficients.restype = None
        lib.gsr_adam_coefficients.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_float)]
        _declared = True
    return lib


class FusedAdam(torch.optim.Adam):
    # ---- fused gradient accumulation (optional) ----------------------------------------------------------------------------------
    # The mapping loop sums 8-10 views into the parameters' gradients before every step. With this switched on, zero_grad() keeps ONE
    # flat zero-filled buffer whose views are the parameters' .grad (mapping_shard.GradBucket.attach) instead of dropping the
    # gradients, and the rasterizer's backward kernels add each view's gradients to it themselves (GSR_BACKWARD_ACCUMULATE,
    # diff_gaussian_rasterization/autograd.py, raw.py): autograd's AccumulateGrad -- six read-modify-write launches per view -- and the
    # per-view gradient allocations disappear. Values are those of autograd's accumulation, bit for bit. The densification code
    # replaces parameters at will: the buffer is rebuilt at the next zero_grad() whenever the parameter list changed.
    def enable_fused_gradient_accumulation(self, on=True):
        self._fused_acc = bool(on)
        self._bucket, self._bucket_key, self._bucket_rest = None, None, []
        return self

    def zero_grad(self, set_to_none=True):
        if not getattr(self, "_fused_acc", False):
            return super().zero_grad(set_to_none)
        key = tuple(id(p) for group in self.param_groups for p in group["params"])
        if key != getattr(self, "_bucket_key", None):          # first call, or densification / an opacity reset replaced parameters
            from mapping_shard import GradBucket, _is_dense
            params, rest = [], []
            for group in self.param_groups:
                for p in group["params"]:
                    ok = p.requires_grad and p.numel() and p.is_cuda and p.dtype == torch.float32 and _is_dense(p)
                    (params if ok else rest).append(p)
            self._bucket = GradBucket(params).attach(fused_accumulate=True) if params else None     # zero-filled on creation
            self._bucket_rest, self._bucket_key = rest, key
        elif self._bucket is not None:
            self._bucket.zero_grads()
        for p in self._bucket_rest:
            p.grad = None

    def _fusable(self, todo):
        if not todo or len(todo) > _MAX_SEGMENTS:
            return False
        for group, p in todo:
            if group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"] or group.get("capturable") or group.get("differentiable"):
                return False
            g = p.grad
            if (not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse
                    or not p.is_contiguous() or not g.is_contiguous()):
                return False
            st = self.state.get(p, None)
            if st:          # moments edited from outside (densification surgery, load_state_dict): the kernel reads them through raw pointers
                for k in ("exp_avg", "exp_avg_sq"):
                    t = st.get(k)
                    if (t is None or not t.is_cuda or t.device != p.device or t.dtype != torch.float32 or not t.is_contiguous()
                            or t.numel() != p.numel()):
                        return False
        return True

    def _todo(self):
        return [(group, p) for group in self.param_groups for p in group["params"] if p.grad is not None]

    def _segments(self, todo, advance):
        segs = (_Segment * len(todo))()
        for k, (group, p) in enumerate(todo):
            st = self.state[p]
            if len(st) == 0:          # lazy state initialisation, as torch.optim.Adam does it
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if advance:
                st["step"] += 1
            b1, b2 = group["betas"]
            s = segs[k]
            s.param, s.grad, s.exp_avg, s.exp_avg_sq = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            s.n, s.lr, s.beta2, s.eps, s.step = p.numel(), float(group["lr"]), float(b2), float(group["eps"]), int(st["step"])
            s.beta1_d, s.beta2_d = float(b1), float(b2)
        return segs

    @torch.no_grad()
    def step(self, closure=None):
        todo = self._todo()
        if not self._fusable(todo):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        segs = self._segments(todo, advance=True)
        dev = todo[0][1].device
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_adam_step(len(todo), segs, _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_adam_step")
        return loss

    # ---- the step inside a captured graph (slam/mapping_graph.py) --------------------------------------------------------------------
    # A hipGraph that contains optimizer.step() is replayed for many iterations, but the bias corrections (and the xyz learning rate) change
    # with every step: the kernel then reads its two step-dependent coefficients per parameter tensor from DEVICE memory
    # (gsr_adam_step_scheduled), which the caller fills per iteration from a precomputed schedule. The arithmetic is step()'s, bit for bit.
    def scheduled_segments(self):
        """The (group, parameter) pairs a scheduled step covers, in segment order; None when the fused kernel cannot take them."""
        todo = self._todo()
        if not self._fusable(todo):
            return None
        for _, p in todo:                       # the moments must exist before a capture (step() creates them lazily)
            if len(self.state[p]) == 0:
                return None
        return todo

    @staticmethod
    def coefficients(lr, betas, step):
        """(lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) as fp32, evaluated by the library exactly as gsr_adam_step evaluates them."""
        lib = _lib()
        out = (C.c_float * 2)()
        lr = C.c_float(float(lr)).value        # step() hands the learning rate over as an fp32 field of gsr_adam_segment: the same rounding here
        lib.gsr_adam_coefficients(lr, float(betas[0]), float(betas[1]), int(step), out)
        return float(out[0]), float(out[1])

    @torch.no_grad()
    def step_scheduled(self, todo, coefficients):
        """One step over `todo` (scheduled_segments()) with the coefficients at the device address `coefficients` ([len(todo), 2] fp32).
        Does NOT advance state["step"]: the caller adds the number of executed iterations afterwards (advance_steps)."""
        segs = self._segments(todo, advance=False)
        dev = todo[0][1].device
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_adam_step_scheduled(len(todo), segs, int(coefficients), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_adam_step_scheduled")

    # ---- the step on one rank's slice of a flat parameter buffer (mapping_shard.ShardedMappingStep, exchange="reduce_scatter") ------------
    @torch.no_grad()
    def step_slice(self, param_bucket, lo, hi):
        """Adam on the elements [lo, hi) of ``param_bucket.flat`` only (the parameters are views of it, their gradients views of a flat
        gradient buffer with the same layout): one fused launch over the pieces of the parameter tensors that fall into the range. Every
        parameter's step count advances (all ranks keep the same counters); moments outside the range are left alone."""
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        segs = (_Segment * _MAX_SEGMENTS)()
        n_seg, dev = 0, None
        for p, off in zip(param_bucket.params, param_bucket.offsets):
            if p.grad is None:
                continue
            group = group_of[id(p)]
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            a, b = max(lo, off), min(hi, off + p.numel())
            if a >= b:
                continue
            if (group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"] or not p.is_cuda or not p.is_contiguous()
                    or not p.grad.is_contiguous() or n_seg >= _MAX_SEGMENTS):
                raise RuntimeError("FusedAdam.step_slice: plain Adam on contiguous float32 device tensors, at most 8 pieces")
            s, shift = segs[n_seg], 4 * (a - off)
            s.param, s.grad = p.data_ptr() + shift, p.grad.data_ptr() + shift
            s.exp_avg, s.exp_avg_sq = st["exp_avg"].data_ptr() + shift, st["exp_avg_sq"].data_ptr() + shift
            b1, b2 = group["betas"]
            s.n, s.lr, s.beta2, s.eps, s.step = b - a, float(group["lr"]), float(b2), float(group["eps"]), int(st["step"])
            s.beta1_d, s.beta2_d = float(b1), float(b2)
            n_seg, dev = n_seg + 1, p.device
        if n_seg == 0:
            return
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_adam_step(n_seg, segs, _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_adam_step (slice)")

    def advance_steps(self, todo, n):
        for _, p in todo:
            self.state[p]["step"] += n


class DeviceCountAdam(torch.optim.Adam):
    """``torch.optim.Adam(..., fused=True, capturable=True)`` -- the step counts are float32 scalars ON THE DEVICE, so a step recorded in a
    hipGraph replays correctly -- whose ``step()`` is two launches of this library (gsr_adam_step_device_count: the counts advanced and the bias
    corrections evaluated by one tiny kernel, then every parameter tensor in one launch) instead of torch's multi-tensor kernels, which give the
    node network's ~25 tensors 25 workgroups (2 x 24 us per step; here ~8). Same state layout as torch's (``state[p] = {"step", "exp_avg",
    "exp_avg_sq"}``): state surgery, ``state_dict`` and the first step (which creates the state) are torch's own. The arithmetic is
    gsr_adam_step's (= torch's single-tensor Adam). Anything it does not cover -- amsgrad, weight decay, maximize, more than 32 tensors, tensors
    that are not contiguous float32 on one device -- goes through torch's step()."""

    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        kw.setdefault("capturable", True)
        super().__init__(params, **kw)
        self._coefficients = None

    def _plan(self):
        todo = [(group, p) for group in self.param_groups for p in group["params"] if p.grad is not None]
        if not todo or len(todo) > _MAX_SEGMENTS:
            return None
        dev = todo[0][1].device
        for group, p in todo:
            if group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"] or group.get("differentiable"):
                return None
            st = self.state.get(p)
            if not st:
                return None                                 # the first step creates the state: torch's
            tensors = (p, p.grad, st.get("exp_avg"), st.get("exp_avg_sq"))
            if any(t is None or not t.is_cuda or t.device != dev or t.dtype != torch.float32 or t.is_sparse or not t.is_contiguous() or t.numel() != p.numel()
                   for t in tensors):
                return None
            step = st.get("step")
            if not torch.is_tensor(step) or not step.is_cuda or step.device != dev or step.dtype != torch.float32 or step.numel() != 1:
                return None
        return todo

    @torch.no_grad()
    def step(self, closure=None):
        todo = self._plan()
        if todo is None:
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        n = len(todo)
        dev = todo[0][1].device
        if self._coefficients is None or self._coefficients.device != dev:
            self._coefficients = torch.zeros((2 * _MAX_SEGMENTS,), dtype=torch.float32, device=dev)
        segs = (_Segment * n)()
        counts = (C.c_void_p * n)()
        for k, (group, p) in enumerate(todo):
            st = self.state[p]
            b1, b2 = group["betas"]
            s = segs[k]
            s.param, s.grad, s.exp_avg, s.exp_avg_sq = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            s.n, s.lr, s.beta2, s.eps, s.step = p.numel(), float(group["lr"]), float(b2), float(group["eps"]), 1
            s.beta1_d, s.beta2_d = float(b1), float(b2)
            counts[k] = st["step"].data_ptr()
        lib = _lib()
        with torch.cuda.device(dev):
            rc = lib.gsr_adam_step_device_count(n, segs, counts, self._coefficients.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_adam_step_device_count")
        return loss
