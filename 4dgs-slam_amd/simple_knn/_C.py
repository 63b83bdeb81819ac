"""ctypes binding for the k-NN entry point of libgs_rasterizer_hip.so -- the counterpart of the reference's pybind
module ``simple_knn._C`` (submodules/simple-knn/ext.cpp, spatial.cu:15-26). No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _rast
from diff_gaussian_rasterization._C import load_library, _err, _require_device

_declared = False


def _lib():
    global _declared
    lib = load_library()
    if not _declared:
        lib.gsr_knn_workspace_size.restype = C.c_size_t
        lib.gsr_knn_workspace_size.argtypes = [C.c_int]
        lib.gsr_knn_mean_dist2.restype = C.c_int
        lib.gsr_knn_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _declared = True
    return lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """float[P]: mean squared distance of every point to its 3 nearest other points (spatial.cu:15-26)."""
    _require_device(points, "points")
    if points.dtype != torch.float32:
        raise RuntimeError(f"points must be float32, got {points.dtype}")
    if _rast._glue is not None:
        with torch.cuda.device(points.device):
            return _rast._glue.dist_cuda2(points, torch.cuda.current_stream(points.device).cuda_stream)
    lib = _lib()
    P = int(points.shape[0])
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)  # spatial.cu:21 (torch::full 0)
    if P == 0:
        return means
    pts = points.contiguous()
    if pts.dtype != torch.float32:
        raise RuntimeError(f"points must be float32, got {pts.dtype}")
    ws = torch.empty(lib.gsr_knn_workspace_size(P), dtype=torch.uint8, device=points.device)
    with torch.cuda.device(points.device):
        rc = lib.gsr_knn_mean_dist2(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(),
                                    torch.cuda.current_stream(points.device).cuda_stream)
    if rc < 0:
        _err(lib, rc, "gsr_knn_mean_dist2")
    return means
