"""Fused mapping loss: the counterpart of ``utils/slam_utils.py:252-364`` (get_loss_mapping,
get_loss_mapping_rgbd) with the same signature. Every pixel mask of the reference is a constant of the keyframe and folds into
two weight images; the weighted L1 itself -- value, the cotangents dL/dI and dL/dD that feed the rasterizer's backward, and
the exposure gradients -- is two HIP kernels (include/slam_losses.h) instead of ~20 torch kernels plus their autograd replay.

There is no CPU implementation here: CPU tensors raise, as everywhere in the product path."""
import collections
import ctypes as C
import weakref

import torch

from diff_gaussian_rasterization import _C

_declared = False
_NATIVE_NODE = __import__("os").environ.get("GSR_NATIVE_AUTOGRAD", "1") != "0"


class _MaskedTerm(C.Structure):      # gsr_masked_l1_term, include/slam_losses.h
    _fields_ = [("image", C.c_void_p), ("target", C.c_void_p), ("mask", C.c_void_p), ("dL_dimage", C.c_void_p)]


def _lib():
    global _declared
    lib = _C.load_library()
    if not _declared:
        vp, f, i = C.c_void_p, C.c_float, C.c_int
        lib.gsr_l1_loss_workspace_size.restype = C.c_size_t
        lib.gsr_l1_loss_forward.restype = i
        lib.gsr_l1_loss_forward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, vp, f, vp, f, vp, vp, vp]
        lib.gsr_l1_loss_backward.restype = i
        lib.gsr_l1_loss_backward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, vp, f, vp, f, vp, vp, vp, vp, vp, vp]
        lib.gsr_masked_l1_forward.restype = i
        lib.gsr_masked_l1_forward.argtypes = [i, C.POINTER(_MaskedTerm), i, i, i, i, f, vp, vp, vp]
        lib.gsr_masked_l1_backward.restype = i
        lib.gsr_masked_l1_backward.argtypes = [i, C.POINTER(_MaskedTerm), i, i, i, i, f, vp, vp]
        lib.gsr_ssim_workspace_size.restype = C.c_size_t
        lib.gsr_ssim_workspace_size.argtypes = [i, i, i]
        lib.gsr_ssim_forward.restype = i
        lib.gsr_ssim_forward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp]
        lib.gsr_ssim_backward.restype = i
        lib.gsr_ssim_backward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, vp]
        lib.gsr_densification_stats.restype = i
        lib.gsr_densification_stats.argtypes = [i, vp, vp, vp, vp, vp, vp]
        _declared = True
    return lib


def _p(t, keep):
    if t is None:
        return None
    _C._require_device(t, "loss operand")
    tc = t.detach()
    tc = tc if tc.dtype == torch.float32 else tc.to(torch.float32)
    tc = tc.contiguous()
    keep.append(tc)
    return tc.data_ptr()


_PLACEHOLDERS = {}


def _placeholder(device):
    """The value of a loss evaluated with compute_value=False: a zero scalar (a defined placeholder, NOT the loss: sums and logs stay finite).
    Outside a stream capture it is a fresh tensor, so a caller's in-place arithmetic (`loss += x`) touches nobody else's value.
    During a capture -- where a fill launch per loss term is what the flag exists to avoid, and the only callers are this repo's graph
    builders (slam/tracking_graph.py, mapping_graph.py, dynamic_graph.py), which never write into a loss value -- it is an alias of ONE
    zero per device: read-only by contract."""
    if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return torch.zeros((), dtype=torch.float32, device=device)
    key = str(device)
    z = _PLACEHOLDERS.get(key)
    if z is None:
        z = _PLACEHOLDERS[key] = torch.zeros((), dtype=torch.float32, device=device)
    return z.detach()


class _WeightedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, alpha, opacity=None, opacity_thr=0.95,
                compute_value=True):
        _C._require_device(image, "image")
        ctx.alpha, ctx.opacity_thr = float(alpha), float(opacity_thr)
        opacity = None if opacity is None else opacity.detach()
        if not compute_value:
            # the caller only back-propagates (a tracking iteration inside a hipGraph): the value's two launches are skipped; the backward
            # kernels need nothing from them (the workspace is their scratch)
            loss = _placeholder(image.device)
            ws = torch.empty((int(_lib().gsr_l1_loss_workspace_size()),), dtype=torch.uint8, device=image.device)
            ctx.save_for_backward(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, ws, opacity)
            return loss
        if _C._glue is not None:     # native host glue (csrc/torch_glue.cpp)
            with torch.cuda.device(image.device):
                loss, ws = _C._glue.l1_loss_forward(image.detach(), depth.detach(), gt_image, gt_depth, w_rgb, w_depth,
                                                    None if exposure_a is None else exposure_a.detach(),
                                                    None if exposure_b is None else exposure_b.detach(), float(alpha), opacity,
                                                    float(opacity_thr), _C._stream(image.device))
            ctx.save_for_backward(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, ws, opacity)
            return loss
        lib = _lib()
        H, W = int(image.shape[-2]), int(image.shape[-1])
        dev = image.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty((int(lib.gsr_l1_loss_workspace_size()),), dtype=torch.uint8, device=dev)
        keep = []
        with torch.cuda.device(dev):
            rc = lib.gsr_l1_loss_forward(W, H, _p(image, keep), _p(depth, keep), _p(gt_image, keep), _p(gt_depth, keep), _p(w_rgb, keep),
                                         _p(w_depth, keep), _p(exposure_a, keep), _p(exposure_b, keep), float(alpha), _p(opacity, keep),
                                         float(opacity_thr), loss.data_ptr(), ws.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_l1_loss_forward")
        ctx.save_for_backward(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, ws, opacity)
        return loss

    @staticmethod
    def backward(ctx, g):
        image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, ws, opacity = ctx.saved_tensors
        if _C._glue is not None:
            with torch.cuda.device(image.device):
                g_image, g_depth, g_exp = _C._glue.l1_loss_backward(
                    image.detach(), depth.detach(), gt_image, gt_depth, w_rgb, w_depth, None if exposure_a is None else exposure_a.detach(),
                    None if exposure_b is None else exposure_b.detach(), ctx.alpha, opacity, ctx.opacity_thr, g, ws, _C._stream(image.device))
            ga = g_exp[0:1].view(exposure_a.shape) if exposure_a is not None else None
            gb = g_exp[1:2].view(exposure_b.shape) if exposure_b is not None else None
            return g_image, g_depth, None, None, None, None, ga, gb, None, None, None, None
        lib = _lib()
        H, W = int(image.shape[-2]), int(image.shape[-1])
        dev = image.device
        g_image, g_depth = torch.empty_like(image, dtype=torch.float32), torch.empty_like(depth, dtype=torch.float32)
        g_exp = torch.empty((2,), dtype=torch.float32, device=dev) if exposure_a is not None else None
        keep = []
        with torch.cuda.device(dev):
            rc = lib.gsr_l1_loss_backward(W, H, _p(image, keep), _p(depth, keep), _p(gt_image, keep), _p(gt_depth, keep), _p(w_rgb, keep),
                                          _p(w_depth, keep), _p(exposure_a, keep), _p(exposure_b, keep), ctx.alpha, _p(opacity, keep),
                                          ctx.opacity_thr, _p(g, keep),
                                          g_image.data_ptr(), g_depth.data_ptr(), g_exp.data_ptr() if g_exp is not None else None,
                                          ws.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_l1_loss_backward")
        ga = g_exp[0:1].view(exposure_a.shape) if exposure_a is not None else None
        gb = g_exp[1:2].view(exposure_b.shape) if exposure_b is not None else None
        return g_image, g_depth, None, None, None, None, ga, gb, None, None, None, None


def weighted_l1_loss(image, depth, gt_image, gt_depth, w_rgb=None, w_depth=None, exposure_a=None, exposure_b=None, alpha=0.95,
                     opacity=None, opacity_depth_threshold=0.95, compute_value=True):
    """alpha * mean(w_rgb |exp(a) image + b - gt_image|) + (1 - alpha) * mean(w_depth |depth - gt_depth|), differentiable in
    image, depth, exposure_a, exposure_b. image [3,H,W], depth [1,H,W] (or [H,W]), weights [H,W] / [1,H,W] or None.
    opacity (the rendered opacity, tracking loss): w_rgb *= opacity, w_depth *= (opacity > opacity_depth_threshold); it is a
    constant weight here -- the rasterizer's backward discards the opacity cotangent in any case.
    compute_value=False: the returned scalar is uninitialised (for callers that only call .backward() on it)."""
    if (exposure_a is None) != (exposure_b is None):
        raise RuntimeError("weighted_l1_loss: give both exposure parameters or neither")
    glue = _C._glue
    if (_NATIVE_NODE and glue is not None and hasattr(glue, "weighted_l1_autograd") and image.is_cuda and image.dtype == torch.float32
            and image.device.index == torch.cuda.current_device()):
        # the same node in C++ (csrc/torch_glue.cpp WeightedL1Node): one call per view and mapping iteration, ~30 us less host time each way
        return glue.weighted_l1_autograd(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, float(alpha), opacity,
                                         float(opacity_depth_threshold), bool(compute_value), _C._stream(image.device))
    return _WeightedL1.apply(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, alpha, opacity, opacity_depth_threshold,
                             bool(compute_value))


class _MaskedL1(torch.autograd.Function):
    """scale * sum_terms mean(|target - image[:C] * mask|) over up to four (image, target, mask) terms: two launches forward, one back."""

    @staticmethod
    def forward(ctx, scale, channels, compute_value, *ops):
        images, targets, masks = ops[0::3], ops[1::3], ops[2::3]
        lib = _lib()
        dev = images[0].device
        Cimg, H, W = (int(v) for v in images[0].shape)
        keep = []
        terms = (_MaskedTerm * len(images))()
        for t, (im, tg, mk) in enumerate(zip(images, targets, masks)):
            if tuple(im.shape) != (Cimg, H, W) or tg.numel() != channels * H * W or mk.numel() != H * W:
                raise RuntimeError(f"masked_l1: term {t}: image {tuple(im.shape)}, target {tuple(tg.shape)}, mask {tuple(mk.shape)}")
            terms[t].image, terms[t].target, terms[t].mask = _p(im, keep), _p(tg, keep), _p(mk, keep)
        if compute_value:
            loss = torch.empty((), dtype=torch.float32, device=dev)
            ws = torch.empty((int(lib.gsr_l1_loss_workspace_size()),), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.gsr_masked_l1_forward(len(images), terms, W, H, int(channels), Cimg, float(scale), loss.data_ptr(), ws.data_ptr(), _C._stream(dev))
            if rc < 0:
                _C._err(lib, rc, "gsr_masked_l1_forward")
        else:               # the caller only back-propagates (a captured mapping iteration): the backward kernel needs nothing from the forward pass
            loss = _placeholder(dev)
        ctx.scale, ctx.channels = float(scale), int(channels)
        ctx.save_for_backward(*keep)
        return loss

    @staticmethod
    def backward(ctx, g):
        ops = ctx.saved_tensors
        lib = _lib()
        n = len(ops) // 3
        dev = ops[0].device
        Cimg, H, W = (int(v) for v in ops[0].shape)
        grads = torch.empty((n, Cimg, H, W), dtype=torch.float32, device=dev)
        terms = (_MaskedTerm * n)()
        for t in range(n):
            terms[t].image, terms[t].target, terms[t].mask = ops[3 * t].data_ptr(), ops[3 * t + 1].data_ptr(), ops[3 * t + 2].data_ptr()
            terms[t].dL_dimage = grads[t].data_ptr()
        keep = []
        with torch.cuda.device(dev):
            rc = lib.gsr_masked_l1_backward(n, terms, W, H, ctx.channels, Cimg, ctx.scale, _p(g, keep), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_masked_l1_backward")
        out = [None, None, None]
        for t in range(n):
            out += [grads[t], None, None]
        return tuple(out)


def masked_l1(scale, terms, channels, compute_value=True):
    """scale * sum over `terms` = [(image [Cimg,H,W], target [channels,H,W], mask [H,W] or [1,H,W]), ...] (at most four) of
    mean(|target - image[:channels] * mask|): the optical-flow terms of the dynamic mapping loop (utils/slam_backend.py:479-509) in one
    forward and one backward call. Differentiable in the images only; targets and masks are constants (fp32, the target already masked)."""
    flat = [t for term in terms for t in term]
    return _MaskedL1.apply(float(scale), int(channels), bool(compute_value), *flat)


# Ground-truth-only constants of a frame (device copy of the depth map + four masks, ~10 MB at 640x480). They live in a BOUNDED cache
# keyed by the viewpoint object, not on the viewpoint: get_loss_tracking runs for every frame, the reference's Camera.clean()
# (utils/camera_utils.py:438-448) knows nothing about an extra attribute, and an unbounded per-frame cache grows by tens of GB over a
# sequence. 24 entries cover the mapping window (8) + the two random keyframes + the frame being tracked with room to spare.
_CONST_CACHE = collections.OrderedDict()
_CONST_CACHE_MAX = 24


def _keyframe_constants(config, viewpoint, device):
    """(gt_image, gt_depth, rgb mask, depth mask, tracking rgb mask, tracking depth mask) on `device` (slam_utils.py:276-284, 65-77).
    gt_depth and the depth masks are None for frames without a depth map (monocular)."""
    grad_mask = getattr(viewpoint, "grad_mask", None)
    key = (id(viewpoint.depth), id(viewpoint.original_image), str(device), float(config["Training"]["rgb_boundary_threshold"]), id(grad_mask))
    ent = _CONST_CACHE.get(id(viewpoint))
    if ent is not None and ent[0]() is viewpoint and ent[1] == key:
        _CONST_CACHE.move_to_end(id(viewpoint))
        return ent[2]
    gt_image = viewpoint.original_image.to(device)
    rgb = (gt_image.sum(dim=0) > config["Training"]["rgb_boundary_threshold"])[None].to(torch.float32)
    gt_depth = dep = t_dep = None
    if viewpoint.depth is not None:
        gt_depth = torch.as_tensor(viewpoint.depth, dtype=torch.float32, device=device)[None]
        dep = ((gt_depth > 0.01) & (gt_depth < 10000.0)).to(torch.float32)
        t_dep = ((gt_depth > 0.01) & (gt_depth < 1000.0)).to(torch.float32)
    t_rgb = rgb * grad_mask.view(*rgb.shape) if grad_mask is not None else None           # tracking only (slam_utils.py:70,118-119)
    data = (gt_image, gt_depth, rgb, dep, t_rgb, t_dep)
    try:
        ref = weakref.ref(viewpoint, lambda _r, k=id(viewpoint): _CONST_CACHE.pop(k, None))
    except TypeError:                                                     # not weak-referenceable (e.g. SimpleNamespace stand-ins): a strong
        ref = (lambda v=viewpoint: v)                                     # reference; the LRU bound below still caps what is kept alive
    _CONST_CACHE[id(viewpoint)] = (ref, key, data, {})        # {}: loss weights derived from `data` (get_loss_mapping)
    while len(_CONST_CACHE) > _CONST_CACHE_MAX:
        _CONST_CACHE.popitem(last=False)
    return data


_DROP_LISTENERS = []          # weak references to callables(viewpoint | None): other per-keyframe stores that follow Camera.clean()


def on_drop_keyframe_constants(method):
    """Register a bound method to be called by drop_keyframe_constants with the same argument (held weakly: a store that dies unregisters)."""
    _DROP_LISTENERS.append(weakref.WeakMethod(method))


def drop_keyframe_constants(viewpoint=None):
    """Forget the cached constants of one viewpoint (or of all): call it where the reference calls Camera.clean(). Stores registered through
    on_drop_keyframe_constants (the graphs' per-keyframe operands, slam/mapping_graph.py) drop their entries of that viewpoint too."""
    if viewpoint is None:
        _CONST_CACHE.clear()
    else:
        _CONST_CACHE.pop(id(viewpoint), None)
    for ref in list(_DROP_LISTENERS):
        fn = ref()
        if fn is None:
            _DROP_LISTENERS.remove(ref)
        else:
            fn(viewpoint)


def mapping_loss_weights(config, viewpoint, gt_image, gt_depth, rm_dynamic=False, mask=None, dynamic=False, base=None):
    """(w_rgb, w_depth) float32 [1,H,W]: the masks and region weights of get_loss_mapping_rgbd (slam_utils.py:276-290,350-360).
    They depend on the keyframe only (ground truth, motion mask, the caller's mask), never on the rendering.
    `base` = precomputed ground-truth-only masks (rgb, depth) as float tensors."""
    shape = gt_depth.shape
    if base is None:
        rgb = (gt_image.sum(dim=0) > config["Training"]["rgb_boundary_threshold"]).view(*shape)
        dep = (gt_depth > 0.01).view(*shape) & (gt_depth < 10000.0).view(*shape)
        w_rgb, w_dep = rgb.to(torch.float32), dep.to(torch.float32)
    else:
        w_rgb, w_dep = base
    motion = getattr(viewpoint, "motion_mask", None)
    if motion is not None and rm_dynamic:
        w_rgb, w_dep = motion.view(*shape) * w_rgb, motion.view(*shape) * w_dep
    if mask is not None and rm_dynamic:
        w_rgb, w_dep = mask.view(*shape) * w_rgb, mask.view(*shape) * w_dep
    if dynamic:
        boost = (mask.view(*shape).bool() | ~motion.view(*shape)) if mask is not None else ~motion.view(*shape)
        scale = torch.where(boost, 2.0, 1.0)
        w_rgb, w_dep = w_rgb * scale, w_dep * scale
    return w_rgb.to(torch.float32), w_dep.to(torch.float32)


def get_loss_mapping(config, image, depth, viewpoint, opacity, initialization=False, alpha=None, rm_dynamic=False, mask=None,
                     dynamic=False, split=False, compute_value=True):
    """utils/slam_utils.py:252-259, same arguments and value. RGB-D, non-split calls (every call of utils/slam_backend.py with
    the shipped configs) run fused; `monocular` and `split=True` are evaluated with the reference's tensor expression.
    compute_value=False (fused path only): the scalar is left uninitialised -- for callers that only back-propagate it."""
    _C._require_device(image, "image")
    gt_image, gt_depth, base_rgb, base_dep, _, _ = _keyframe_constants(config, viewpoint, image.device)
    exposure = (None, None) if initialization else (viewpoint.exposure_a, viewpoint.exposure_b)
    if config["Training"]["monocular"]:                                           # never touches the depth map (there is none)
        image_ab = image if initialization else torch.exp(exposure[0]) * image + exposure[1]
        return torch.abs(image_ab * base_rgb - gt_image * base_rgb).mean()       # get_loss_mapping_rgb, :262-272
    if alpha is None:
        alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    if split:                                                                     # :292-303
        image_ab = image if initialization else torch.exp(exposure[0]) * image + exposure[1]
        w_rgb, w_dep = mapping_loss_weights(config, viewpoint, gt_image, gt_depth, rm_dynamic, mask, False)
        mm = viewpoint.motion_mask.view(*depth.shape)
        part = lambda sel: (alpha * torch.abs(sel * w_rgb * image_ab - sel * w_rgb * gt_image).mean()
                            + (1 - alpha) * torch.abs(sel * w_dep * depth - sel * w_dep * gt_depth).mean())
        l_static, l_dynamic = part(mm), part(~mm)
        return (l_static, 2 * l_dynamic) if dynamic else (l_static, l_dynamic)
    w_rgb, w_dep = _cached_mapping_weights(config, viewpoint, gt_image, gt_depth, base_rgb, base_dep, rm_dynamic, mask, dynamic)
    return weighted_l1_loss(image, depth, gt_image, gt_depth, w_rgb, w_dep, exposure[0], exposure[1], alpha, compute_value=compute_value)


def _cached_mapping_weights(config, viewpoint, gt_image, gt_depth, base_rgb, base_dep, rm_dynamic, mask, dynamic):
    """The weights are constants of (keyframe, flags, masks): formed once, not per call (2-4 image-sized launches per view and iteration)."""
    motion = getattr(viewpoint, "motion_mask", None)
    wkey = (bool(rm_dynamic), bool(dynamic), id(mask), getattr(mask, "_version", None), id(motion), getattr(motion, "_version", None))
    derived = _CONST_CACHE[id(viewpoint)][3]
    hit = derived.get(wkey)
    if hit is None:
        if len(derived) >= 8:
            derived.clear()
        hit = derived[wkey] = mapping_loss_weights(config, viewpoint, gt_image, gt_depth, rm_dynamic, mask, dynamic, base=(base_rgb, base_dep)) + (mask, motion)
    return hit[0], hit[1]


def mapping_loss_operands(config, viewpoint, device, rm_dynamic=False, mask=None, dynamic=False):
    """(gt_image [3,H,W], gt_depth [1,H,W], w_rgb [1,H,W], w_depth [1,H,W], alpha): the constant operands the fused get_loss_mapping call of
    this keyframe hands to weighted_l1_loss -- the SAME tensors (values and, while the constants' cache holds the keyframe, storage). The
    graph-captured mapping iteration (slam/mapping_graph.py) keeps them per keyframe and calls weighted_l1_loss itself. RGB-D only."""
    if config["Training"]["monocular"]:
        raise RuntimeError("mapping_loss_operands: RGB-D keyframes only")
    gt_image, gt_depth, base_rgb, base_dep, _, _ = _keyframe_constants(config, viewpoint, device)
    w_rgb, w_dep = _cached_mapping_weights(config, viewpoint, gt_image, gt_depth, base_rgb, base_dep, rm_dynamic, mask, dynamic)
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    return gt_image, gt_depth, w_rgb, w_dep, alpha


def tracking_loss_weights(config, viewpoint, gt_image, gt_depth, rm_dynamic=False, mask=None, base=None):
    """(w_rgb, w_depth) float32 [1,H,W] of get_loss_tracking_rgbd WITHOUT the rendered-opacity factors (slam_utils.py:65-77,118-135):
    rgb: boundary threshold x grad_mask [x motion mask if rm_dynamic and uid > 0] [x mask]; depth: 0.01 < d < 1000 [x motion] [x mask].
    `base` = the two ground-truth-only products, precomputed."""
    shape = gt_depth.shape
    if base is None:
        w_rgb = (gt_image.sum(dim=0) > config["Training"]["rgb_boundary_threshold"]).view(*shape) * viewpoint.grad_mask.view(*shape)
        w_dep = (gt_depth > 0.01).view(*shape) & (gt_depth < 1000.0).view(*shape)
    else:
        w_rgb, w_dep = base
    motion = getattr(viewpoint, "motion_mask", None)
    if motion is not None and rm_dynamic and viewpoint.uid > 0:
        w_rgb, w_dep = motion.view(*shape) * w_rgb, motion.view(*shape) * w_dep
    if mask is not None:
        w_rgb, w_dep = mask.view(*shape) * w_rgb, mask.view(*shape) * w_dep
    return w_rgb.to(torch.float32), w_dep.to(torch.float32)


def get_loss_tracking(config, image, depth, opacity, viewpoint, initialization=False, rm_dynamic=False, mask=None, save_img=False):
    """utils/slam_utils.py:57-61, same arguments and value (RGB-D: fused; monocular: the reference's tensor expression; save_img
    is a debugging aid of the reference and is ignored). The exposure is always applied (:58)."""
    _C._require_device(image, "image")
    gt_image, gt_depth, _, _, t_rgb, t_dep = _keyframe_constants(config, viewpoint, image.device)
    if t_rgb is None:
        raise RuntimeError("get_loss_tracking: viewpoint.grad_mask is not set (call compute_grad_mask first, utils/slam_frontend.py:678)")
    if config["Training"]["monocular"]:                                           # get_loss_tracking_rgb, :64-105: no depth involved
        w_rgb = t_rgb if mask is None else mask.view(*t_rgb.shape) * t_rgb
        motion = getattr(viewpoint, "motion_mask", None)
        if motion is not None and rm_dynamic and viewpoint.uid > 0:
            w_rgb = motion.view(*t_rgb.shape) * w_rgb
        image_ab = torch.exp(viewpoint.exposure_a) * image + viewpoint.exposure_b
        return (opacity * torch.abs(image_ab * w_rgb - gt_image * w_rgb)).mean()
    w_rgb, w_dep = tracking_loss_weights(config, viewpoint, gt_image, gt_depth, rm_dynamic, mask, base=(t_rgb, t_dep))
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    return weighted_l1_loss(image, depth, gt_image, gt_depth, w_rgb, w_dep, viewpoint.exposure_a, viewpoint.exposure_b, alpha,
                            opacity=opacity, opacity_depth_threshold=0.95)


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, mask):
        _C._require_device(img1, "img1")
        ctx.shape = img1.shape
        if _C._glue is not None:
            with torch.cuda.device(img1.device):
                out, ws = _C._glue.ssim_forward(img1.detach(), img2, mask, _C._stream(img1.device))
            ctx.save_for_backward(img1, img2, mask, ws)
            ctx.native = True
            return out
        ctx.native = False
        lib = _lib()
        Cn, H, W = int(img1.shape[-3]), int(img1.shape[-2]), int(img1.shape[-1])
        dev = img1.device
        keep = []
        a, b = _p(img1, keep), _p(img2, keep)
        m8 = None
        if mask is not None:
            m8 = mask.to(torch.uint8).contiguous().view(-1)
            if m8.numel() != H * W:
                raise RuntimeError("ssim: mask must have H*W elements")
        out = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty((int(lib.gsr_ssim_workspace_size(W, H, Cn)),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gsr_ssim_forward(W, H, Cn, a, b, m8.data_ptr() if m8 is not None else None, out.data_ptr(), ws.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_ssim_forward")
        ctx.save_for_backward(keep[0], keep[1], m8, ws)
        ctx.shape = img1.shape
        return out

    @staticmethod
    def backward(ctx, g):
        img1, img2, m8, ws = ctx.saved_tensors
        if ctx.native:
            with torch.cuda.device(img1.device):
                return _C._glue.ssim_backward(img1.detach(), img2, m8, g, ws, _C._stream(img1.device)), None, None
        lib = _lib()
        Cn, H, W = int(ctx.shape[-3]), int(ctx.shape[-2]), int(ctx.shape[-1])
        dev = img1.device
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
        keep = []
        with torch.cuda.device(dev):
            rc = lib.gsr_ssim_backward(W, H, Cn, img1.data_ptr(), img2.data_ptr(), m8.data_ptr() if m8 is not None else None, _p(g, keep),
                                       grad.data_ptr(), ws.data_ptr(), _C._stream(dev))
        if rc < 0:
            _C._err(lib, rc, "gsr_ssim_backward")
        return grad, None, None


def ssim(img1, img2, window_size=11, size_average=True, mask=None):
    """gaussian_splatting/utils/loss_utils.py:63-76, same arguments and value; differentiable in img1 (the rendering; the reference's
    callers pass the ground truth as img2, utils/slam_backend.py:636,824-832). Fused for the only form the reference uses
    (window 11, size_average=True); anything else raises."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the form utils/slam_backend.py uses) is implemented")
    if img2.requires_grad:
        raise RuntimeError("ssim: img2 is treated as the ground truth (no gradient)")
    return _Ssim.apply(img1, img2, mask)


@torch.no_grad()
def add_densification_stats(gaussians, viewspace_point_tensor, radii):
    """One launch for what utils/slam_backend.py:712-720 and GaussianModel.add_densification_stats (gaussian_model.py:973-977) do per
    rendered view with boolean-mask indexing (three host synchronisations): for radii > 0 update gaussians.max_radii2D,
    gaussians.xyz_gradient_accum and gaussians.denom in place. `radii` and `viewspace_point_tensor` come from render()."""
    lib = _lib()
    g = viewspace_point_tensor.grad
    _C._require_device(g, "viewspace_point_tensor.grad")
    P = int(radii.shape[0])
    for name in ("max_radii2D", "xyz_gradient_accum", "denom"):
        t = getattr(gaussians, name)
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P:
            raise RuntimeError(f"add_densification_stats: gaussians.{name} must be a contiguous float32 tensor with one value per Gaussian")
    g = g if g.is_contiguous() else g.contiguous()
    r = radii if (radii.dtype == torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
    with torch.cuda.device(g.device):
        rc = lib.gsr_densification_stats(P, r.data_ptr(), g.data_ptr(), gaussians.max_radii2D.data_ptr(),
                                         gaussians.xyz_gradient_accum.data_ptr(), gaussians.denom.data_ptr(), _C._stream(g.device))
    if rc < 0:
        _C._err(lib, rc, "gsr_densification_stats")
