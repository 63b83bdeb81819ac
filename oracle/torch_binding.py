"""oracle/torch_binding.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU stand-in for the ``diff_gaussian_rasterization`` package whose forward/backward run the C oracle
(oracle/gs_oracle.c). Used (a) in this container to run the reference's UNMODIFIED Python wrapper
(gaussian_splatting/gaussian_renderer) end to end and record golden vectors (tests/golden/make_golden.py), and
(b) by the CPU-only tests that need a rasterizer behind the wrapper / the gloo sharding test. Never imported by
anything under 4dgs-slam_amd/.
"""
from typing import NamedTuple

import numpy as np
import torch
import torch.nn as nn

from . import rasterize_backward, rasterize_forward, mark_visible


class GaussianRasterizationSettings(NamedTuple):
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


def _np(t):
    return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, rs):
        out, st = rasterize_forward(
            bg=_np(rs.bg), means3D=_np(means3D) if means3D.numel() else np.zeros((0, 3), np.float32), opacities=_np(opacities),
            viewmatrix=_np(rs.viewmatrix), projmatrix=_np(rs.projmatrix), campos=_np(rs.campos), tanfovx=rs.tanfovx,
            tanfovy=rs.tanfovy, image_height=rs.image_height, image_width=rs.image_width, shs=_np(sh),
            colors_precomp=_np(colors_precomp), scales=_np(scales), rotations=_np(rotations), cov3D_precomp=_np(cov3Ds_precomp),
            scale_modifier=rs.scale_modifier, sh_degree=rs.sh_degree, prefiltered=rs.prefiltered)
        ctx.st, ctx.rs = st, rs
        ctx.P, ctx.M = means3D.shape[0], (sh.shape[1] if sh.numel() else 0)
        T = lambda a: torch.from_numpy(a)
        return T(out["color"]), T(out["radii"]), T(out["depth"]), T(out["opacity"]), T(out["n_touched"])

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_opacity, g_ntouched):
        rs = ctx.rs
        H, W = rs.image_height, rs.image_width
        gc = np.zeros((3, H, W), np.float32) if g_color is None else g_color.numpy()
        gd = np.zeros((1, H, W), np.float32) if g_depth is None else g_depth.numpy()
        g = rasterize_backward(ctx.st, projmatrix_raw=_np(rs.projmatrix_raw), dL_dcolor=gc, dL_ddepth=gd, P=ctx.P, M=ctx.M)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        tau = g["dL_dtau"].sum(0)
        return (T(g["dL_dmeans3D"]), T(g["dL_dmeans2D"]), T(g["dL_dsh"]), T(g["dL_dcolors"]), T(g["dL_dopacity"]), T(g["dL_dscales"]),
                T(g["dL_drotations"]), T(g["dL_dcov3D"]), T(tau[3:].reshape(1, 3)), T(tau[:3].reshape(1, 3)), None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        rs = self.raster_settings
        return torch.from_numpy(mark_visible(_np(positions), _np(rs.viewmatrix), _np(rs.projmatrix)))

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = lambda t: torch.Tensor([]) if t is None else t
        return _OracleRasterize.apply(means3D, means2D, e(shs), e(colors_precomp), opacities, e(scales), e(rotations),
                                      e(cov3D_precomp), e(theta), e(rho), self.raster_settings)
