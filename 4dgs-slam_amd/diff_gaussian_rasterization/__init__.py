"""Drop-in replacement for the reference's ``diff_gaussian_rasterization`` package
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py), backed by the
MI355X-native HIP library (include/gs_rasterizer.h) instead of the CUDA extension.

Same public names, signatures, argument order, return order and error messages:
``GaussianRasterizationSettings`` (:173-186), ``GaussianRasterizer`` (:188-244), ``rasterize_gaussians`` (:21-46),
``_RasterizeGaussians`` (:48-171, in autograd.py here). Put ``4dgs-slam_amd/`` on ``PYTHONPATH`` and the reference's
``gaussian_splatting/gaussian_renderer`` imports this module unchanged (gaussian_renderer/__init__.py:15-18).
"""
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _C
from .autograd import ACCUMULATE_ATTR, _RasterizeGaussians

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians"]

_MSG_COLOR = "Please provide excatly one of either SHs or precomputed colors!"                  # reference :209 (sic)
_MSG_COV = "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"  # reference :212


class GaussianRasterizationSettings(NamedTuple):
    """13 fields, order and types of the reference (:173-186). Matrices use the row-vector convention of the
    callers: viewmatrix = W2C.T, projmatrix_raw = P.T, projmatrix = W2C.T @ P.T."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_NATIVE_NODE = os.environ.get("GSR_NATIVE_AUTOGRAD", "1") != "0"


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    rs = raster_settings
    glue = _C._glue
    # The same autograd node in C++ (csrc/torch_glue.cpp RasterizeNode) when the native glue is loaded: ~75 us less host time per forward +
    # backward. The Python node below stays the reference path and takes everything the C++ one does not: debug snapshots, fused
    # gradient accumulation into marked parameters (mapping_shard.GradBucket.attach), the ctypes binding, tensors on another device
    # than the current one.
    if (_NATIVE_NODE and glue is not None and hasattr(glue, "rasterize_autograd") and not rs.debug and means3D.is_cuda
            and means3D.device.index == torch.cuda.current_device() and means3D.ndim == 2 and means3D.shape[-1] == 3
            and not any(getattr(p, ACCUMULATE_ATTR, False) for p in (means3D, sh, opacities, scales, rotations))):
        return tuple(glue.rasterize_autograd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, rs.bg,
                                             float(rs.scale_modifier), rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, float(rs.tanfovx), float(rs.tanfovy),
                                             int(rs.image_height), int(rs.image_width), int(rs.sh_degree), rs.campos, bool(rs.prefiltered),
                                             _C._stream(means3D.device)))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     theta, rho, raster_settings)


def _or_empty(t: Optional[torch.Tensor]) -> torch.Tensor:
    # None -> empty CPU tensor == nullptr at the C boundary (reference :214-228)
    return torch.Tensor([]) if t is None else t


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: which positions pass the near-plane test of this camera (reference :193-202)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception(_MSG_COLOR)
        has_sr, any_sr = (scales is not None and rotations is not None), (scales is not None or rotations is not None)
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception(_MSG_COV)
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), opacities, _or_empty(scales),
                                   _or_empty(rotations), _or_empty(cov3D_precomp), _or_empty(theta), _or_empty(rho),
                                   self.raster_settings)
