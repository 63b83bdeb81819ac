"""ctypes binding of libgs_rasterizer_hip.so -- the counterpart of the reference's pybind module
``diff_gaussian_rasterization._C`` (submodules/diff-gaussian-rasterization/ext.cpp:15-19,
rasterize_points.{h,cu}).

Same three entry points, same argument order, same return tuples:
  rasterize_gaussians(...)          -> (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, opacity, n_touched)
  rasterize_gaussians_backward(...) -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau)
  mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

PyTorch is plumbing here: it owns the device memory and the HIP stream; all arithmetic happens in
hand-written HIP kernels behind the C ABI of include/gs_rasterizer.h. There is NO CPU fallback:
tensors must live on a HIP device and the shared library must be present, otherwise this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# GSR_EXACT_MATH=1: the exact-math parity build of the same library (csrc/build.sh --exact); it is driven through ctypes only (the
# native glue is linked against the product library)
EXACT_MATH = os.environ.get("GSR_EXACT_MATH", "0") not in ("", "0")
LIB_PATH = os.environ.get("GSR_LIB", os.path.join(_PKG_ROOT, "libgs_rasterizer_hip_exact.so" if EXACT_MATH else "libgs_rasterizer_hip.so"))
NUM_CHANNELS = 3  # cuda_rasterizer/config.h:15

_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_lib = None

# Two interchangeable host bindings of the same C ABI:
#   "native": csrc/torch_glue.cpp compiled into _glue*.so (tensor allocation + pointer extraction in C++, ~10x less host time)
#   "ctypes": the pure-Python marshalling below (always available once the HIP library is built)
# GSR_GLUE=ctypes|native forces one; by default the native glue is used when it has been built.
_glue = None
_glue_error = None
if os.environ.get("GSR_GLUE", "native") != "ctypes" and not EXACT_MATH and "GSR_LIB" not in os.environ:
    try:
        from . import _glue  # type: ignore
    except Exception as _e:  # not built (or built for another torch): fall back to ctypes, loudly only if it was requested
        _glue, _glue_error = None, _e
        if os.environ.get("GSR_GLUE") == "native":
            raise ImportError(f"GSR_GLUE=native but the native glue cannot be imported: {_e}") from _e


def binding() -> str:
    return "native" if _glue is not None else "ctypes"


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def load_library():
    """dlopen the HIP library (no GPU needed just to load it) and declare the C signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            f"(python __graft_entry__.py build, or 4dgs-slam_amd/csrc/build.sh). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, f, i = C.c_void_p, C.c_float, C.c_int
    lib.gsr_forward.restype = i
    lib.gsr_forward.argtypes = [_ALLOC_FN, vp, _ALLOC_FN, vp, _ALLOC_FN, vp, i, i, i, vp, i, i,
                                vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, i, vp, vp, vp, vp, vp, i, vp]
    lib.gsr_backward.restype = i
    lib.gsr_backward.argtypes = [i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, vp, f, f, vp,
                                 vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]
    lib.gsr_backward_fused.restype = i
    lib.gsr_backward_fused.argtypes = lib.gsr_backward.argtypes[:-2] + [vp, i, vp]
    lib.gsr_mark_visible.restype = i
    lib.gsr_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_version.restype = C.c_char_p
    lib.gsr_geometry_buffer_size.restype = C.c_size_t
    lib.gsr_geometry_buffer_size.argtypes = [i]
    lib.gsr_image_buffer_size.restype = C.c_size_t
    lib.gsr_image_buffer_size.argtypes = [i, i, i]
    lib.gsr_binning_buffer_size.restype = C.c_size_t
    lib.gsr_binning_buffer_size.argtypes = [i]
    lib.gsr_debug_read_state.restype = i
    lib.gsr_debug_read_state.argtypes = [i, i, i, i] + [vp] * 16
    lib.gsr_profile_enable.argtypes = [i]
    lib.gsr_profile_read.restype = i
    lib.gsr_profile_read.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), i]
    lib.gsr_profile_reset.restype = None
    lib.gsr_set_option.restype = i
    lib.gsr_set_option.argtypes = [C.c_char_p, i]
    lib.gsr_forward_status.restype = i
    lib.gsr_forward_status.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.gsr_forward_status_views.restype = i
    lib.gsr_forward_status_views.argtypes = [C.POINTER(C.c_uint)]
    _lib = lib
    return lib


def forward_status_views() -> int:
    """The sticky overflow counter summed over the single-view slot and every view slot of the multi-view entry point (never blocks)."""
    lib = load_library()
    a = C.c_uint(0)
    lib.gsr_forward_status_views(C.byref(a))
    return int(a.value)


def set_option(name: str, value: int = -1) -> int:
    """gsr_set_option (include/gs_rasterizer.h): "speculate" | "lazy" | "mailbox" | "cap_margin_permille"; returns the previous value
    (value < 0: query)."""
    lib = load_library()
    rc = lib.gsr_set_option(name.encode(), int(value))
    if rc < 0:
        _err(lib, rc, "gsr_set_option")
    return rc


def debug_view_slots(max_slots: int = 8):
    """Per capacity slot of this thread (0 = single-view calls, then the view slots of the multi-view entry point): what the GPU last
    reported -- num_rendered, flags, R_alloc, longest tile list, sequence number, sticky overflow count -- and the estimates the next
    speculative layout starts from (gsr_debug_view_slots; never blocks)."""
    lib = load_library()
    buf = (C.c_uint * (8 * max_slots))()
    lib.gsr_debug_view_slots.argtypes = [C.POINTER(C.c_uint), C.c_int]
    n = lib.gsr_debug_view_slots(buf, max_slots)
    keys = ("num_rendered", "flags", "R_alloc", "longest_tile", "seq", "overflows", "estimate_R_alloc", "estimate_longest_tile")
    return [dict(zip(keys, (int(buf[8 * k + j]) for j in range(8)))) for k in range(max(0, n))]


def forward_status():
    """(overflow_count, last_num_rendered) of this thread's forward passes; never blocks (gsr_forward_status)."""
    lib = load_library()
    a, b = C.c_uint(0), C.c_uint(0)
    lib.gsr_forward_status(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def _err(lib, code, what):
    msg = lib.gsr_last_error().decode(errors="replace")
    raise RuntimeError(f"{what} failed (code {code}): {msg}")


def _require_device(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on '{t.device}': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); "
            f"there is no CPU fallback in the product path."
        )


def _ptr(t):
    """data pointer of a float32/int32 tensor made contiguous (rasterize_points.cu:98-118); empty -> NULL."""
    if t is None or t.numel() == 0:
        return None, None
    tc = t.contiguous()
    return tc.data_ptr(), tc


class _Arena:
    """The resizeFunctional of rasterize_points.cu:27-33: a growable uint8 device tensor handed to the C core."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.tensor.data_ptr()


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width,
                        sh, degree, campos, prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-122): same 20 arguments, same 9-tuple."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    _require_device(means3D, "means3D")
    if _glue is not None:
        with torch.cuda.device(means3D.device):
            return _glue.rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp,
                                             viewmatrix, projmatrix, projmatrix_raw, float(tan_fovx), float(tan_fovy), int(image_height),
                                             int(image_width), sh, int(degree), campos, bool(prefiltered), bool(debug), _stream(means3D.device))
    lib = load_library()
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    # one float and one int allocation instead of five (allocator round trips are host time the GPU spends idle)
    img = torch.empty((NUM_CHANNELS + 2, H, W), dtype=torch.float32, device=dev)
    out_color, out_depth, out_opacity = img[:NUM_CHANNELS], img[NUM_CHANNELS:NUM_CHANNELS + 1], img[NUM_CHANNELS + 1:]
    ints = torch.empty((2, P), dtype=torch.int32, device=dev)
    radii, n_touched = ints[0], ints[1]
    geom, binning, imgbuf = _Arena(dev), _Arena(dev), _Arena(dev)
    if P == 0:  # rasterize_points.cu:85: nothing is launched, outputs stay zero
        img.zero_()
        return 0, out_color, radii, geom.tensor, binning.tensor, imgbuf.tensor, out_depth, out_opacity, n_touched
    M = int(sh.shape[1]) if sh.numel() != 0 else 0  # rasterize_points.cu:87-91
    keep = []

    def p(t, name=None):
        if t is not None and t.numel() != 0:
            _require_device(t, name or "tensor")
            if t.dtype != torch.float32:
                raise RuntimeError(f"{name} must be float32, got {t.dtype}")
        ptr, tc = _ptr(t)
        keep.append(tc)
        return ptr

    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.gsr_forward(
            geom.cb, None, binning.cb, None, imgbuf.cb, None, P, int(degree), M,
            p(background, "bg"), W, H, p(means3D, "means3D"), p(sh, "shs"), p(colors, "colors_precomp"), p(opacity, "opacities"),
            p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
            p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
            float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
            out_color.data_ptr(), out_depth.data_ptr(), out_opacity.data_ptr(), radii.data_ptr(), n_touched.data_ptr(),
            int(bool(debug)), stream)
    if rc < 0:
        _err(lib, rc, "gsr_forward")
    return rc, out_color, radii, geom.tensor, binning.tensor, imgbuf.tensor, out_depth, out_opacity, n_touched


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depths,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:124-211): same 23 arguments, same 9-tuple."""
    return rasterize_gaussians_backward_fused(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                              viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color,
                                              dL_dout_depths, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug)[:9]


def rasterize_gaussians_backward_fused(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                       viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depths,
                                       sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, lean=False, accumulate_into=None):
    """Same as rasterize_gaussians_backward plus a 10th result: dL_dtau summed over the Gaussians, float32[6] = [rho, theta]
    (the reduction the reference's autograd Function does with torch.sum, __init__.py:152-154, fused into the kernels).
    lean=True: gradients that only feed other gradients inside the kernel are neither allocated nor written and come back
    as empty tensors -- dL_dtau [P,6] always, dL_dcolors when colours come from SH, dL_dcov3D when it comes from scales and
    rotations (plus the never-returned dL_dconic / dL_ddepth): 80 of the 148 bytes the kernel stores per Gaussian.
    accumulate_into = (g_means3D, g_sh, g_opacity, g_scales, g_rotations): the caller's gradient buffers (fp32, contiguous); this
    view's parameter gradients are ADDED to them by the kernels (GSR_BACKWARD_ACCUMULATE, include/gs_rasterizer.h) and the same
    tensors come back in the result tuple."""
    _require_device(means3D, "means3D")
    accumulate_into = list(accumulate_into) if accumulate_into is not None else []
    if _glue is not None:
        with torch.cuda.device(means3D.device):
            return _glue.rasterize_gaussians_backward_fused(background, means3D, radii, colors, scales, rotations, float(scale_modifier),
                                                            cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, float(tan_fovx),
                                                            float(tan_fovy), dL_dout_color, dL_dout_depths, sh, int(degree), campos,
                                                            geomBuffer, int(R), binningBuffer, imageBuffer, bool(debug), bool(lean),
                                                            _stream(means3D.device), accumulate_into)
    lib = load_library()
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    M = int(sh.shape[1]) if sh.numel() != 0 else 0
    # The kernels write every element, so one torch.empty carved into views replaces the eleven torch.zeros of
    # rasterize_points.cu:160-170 (eleven allocator calls + eleven memsets of host/GPU time per backward).
    # Layout: the five tensors that become the Gaussian parameters' .grad come first, back to back (means3D, sh, opacity,
    # scales, rotations), so mapping_shard.GradBucket can all-reduce them in place as one flat range.
    sh_in, cov_in = M > 0 and colors.numel() == 0, cov3D_precomp.numel() != 0
    acc = len(accumulate_into) == 5
    if accumulate_into and not acc:
        raise ValueError("accumulate_into: five gradient buffers or none")
    pw = [3, 3 * M, 1, 3, 4]
    if acc:
        if not (sh_in and not cov_in):
            raise ValueError("accumulate_into needs the SH + scales / rotations inputs")
        for t_, w_ in zip(accumulate_into, pw):
            if not (t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == P * w_):
                raise ValueError(f"accumulate_into: contiguous fp32 device tensor of {P * w_} elements expected")
    widths = [0 if acc else w_ for w_ in pw] + [3, 0 if (lean and sh_in) else NUM_CHANNELS, 0 if lean else 1, 0 if lean else 4,
                                                 0 if (lean and not cov_in) else 6, 0 if lean else 6]
    flat = (torch.zeros if P == 0 else torch.empty)((P * sum(widths) + 6,), dtype=torch.float32, device=dev)
    views, o = [], 0
    for w_ in widths:
        views.append(flat[o:o + P * w_])
        o += P * w_
    if acc:
        views[:5] = accumulate_into
    dL_dmeans3D, dL_dsh, dL_dopacity = views[0].view(P, 3), views[1].view(P, M, 3), views[2].view(P, 1)
    dL_dscales, dL_drotations, dL_dmeans2D = views[3].view(P, 3), views[4].view(P, 4), views[5].view(P, 3)
    shaped = lambda i, *shape: views[i].view(*shape) if widths[i] else views[i]     # skipped ones stay empty
    dL_dcolors, dL_ddepths, dL_dconic = shaped(6, P, NUM_CHANNELS), shaped(7, P, 1), shaped(8, P, 2, 2)
    dL_dcov3D, dL_dtau = shaped(9, P, 6), shaped(10, P, 6)
    optr = lambda t: t.data_ptr() if t.numel() else None
    tau_sum = flat[o:o + 6]
    if P != 0:
        keep = []

        def p(t, name=None):
            if t is not None and t.numel() != 0:
                _require_device(t, name or "tensor")
            ptr, tc = _ptr(t)
            keep.append(tc)
            return ptr

        sh_path = M > 0 and colors.numel() == 0
        if M > 0 and not sh_path:
            dL_dsh.zero_()  # colours were precomputed: the SH branch is not taken (backward.cu:533)
        gc = dL_dout_color if dL_dout_color.dtype == torch.float32 else dL_dout_color.to(torch.float32)
        gd = dL_dout_depths if dL_dout_depths.dtype == torch.float32 else dL_dout_depths.to(torch.float32)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.gsr_backward_fused(
                P, int(degree), M, int(R), p(background, "bg"), W, H, p(means3D, "means3D"), p(sh, "shs"), p(colors, "colors_precomp"),
                p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(projmatrix_raw, "projmatrix_raw"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy), p(radii, "radii"),
                geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(),
                p(gc, "dL_dout_color"), p(gd, "dL_dout_depth"),
                dL_dmeans2D.data_ptr(), optr(dL_dconic), dL_dopacity.data_ptr(), optr(dL_dcolors), optr(dL_ddepths),
                dL_dmeans3D.data_ptr(), optr(dL_dcov3D), dL_dsh.data_ptr() if sh_path else None,
                dL_dscales.data_ptr(), dL_drotations.data_ptr(), optr(dL_dtau), tau_sum.data_ptr(), int(bool(debug)) | (2 if acc else 0), stream)
        if rc < 0:
            _err(lib, rc, "gsr_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau, tau_sum


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:213-232)."""
    _require_device(means3D, "means3D")
    if _glue is not None:
        with torch.cuda.device(means3D.device):
            return _glue.mark_visible(means3D, viewmatrix, projmatrix, _stream(means3D.device))
    lib = load_library()
    P = int(means3D.shape[0])
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m, v, pr = means3D.contiguous(), viewmatrix.contiguous(), projmatrix.contiguous()
        with torch.cuda.device(means3D.device):
            rc = lib.gsr_mark_visible(P, m.data_ptr(), v.data_ptr(), pr.data_ptr(), present.data_ptr(),
                                      torch.cuda.current_stream(means3D.device).cuda_stream)
        if rc < 0:
            _err(lib, rc, "gsr_mark_visible")
    return present


# ---- extras beyond the reference module (tests / bench) ---------------------------------------------------------
def debug_read_state(P, R, W, H, geomBuffer, binningBuffer, imageBuffer):
    """Intermediate state as numpy arrays, in the reference's GeometryState/ImageState/BinningState terms."""
    import numpy as np
    lib = load_library()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(
        depths=np.zeros(P, np.float32), means2D=np.zeros((P, 2), np.float32), conic_opacity=np.zeros((P, 4), np.float32),
        rgb=np.zeros((P, 3), np.float32), cov3D=np.zeros((P, 6), np.float32), clamped=np.zeros((P, 3), np.uint8),
        tiles_touched=np.zeros(P, np.uint32), point_offsets=np.zeros(P, np.uint32), final_T=np.zeros((H, W), np.float32),
        n_contrib=np.zeros((H, W), np.uint32), ranges=np.zeros((T, 2), np.uint32), point_list=np.zeros(max(R, 0), np.uint32),
    )
    order = ["depths", "means2D", "conic_opacity", "rgb", "cov3D", "clamped", "tiles_touched", "point_offsets",
             "final_T", "n_contrib", "ranges", "point_list"]
    stream = torch.cuda.current_stream(geomBuffer.device).cuda_stream
    rc = lib.gsr_debug_read_state(P, R, W, H, geomBuffer.data_ptr(), binningBuffer.data_ptr() if binningBuffer.numel() else None,
                                  imageBuffer.data_ptr(), *[out[k].ctypes.data_as(C.c_void_p) for k in order], stream)
    if rc < 0:
        _err(lib, rc, "gsr_debug_read_state")
    return out


KERNEL_IDS = {"preprocess_fwd": 0, "scan": 1, "scatter_instances": 2, "sort_tiles": 3, "render_fwd": 4, "render_bwd": 5,
              "geometry_bwd": 6}


def profile_enable(kernels=True):
    """Time kernels with HIP events on the launch stream. True = all, False/None = off, or an iterable of names."""
    if kernels is True:
        mask = -1
    elif not kernels:
        mask = 0
    else:
        mask = 0
        for k in kernels:
            mask |= 1 << KERNEL_IDS[k]
    load_library().gsr_profile_enable(int(mask))


def profile_reset():
    load_library().gsr_profile_reset()


def profile_read():
    lib = load_library()
    cap = 16
    names = (C.c_char_p * cap)()
    ms = (C.c_float * cap)()
    calls = (C.c_int * cap)()
    n = lib.gsr_profile_read(names, ms, calls, cap)
    return {names[i].decode(): (float(ms[i]), int(calls[i])) for i in range(n)}

