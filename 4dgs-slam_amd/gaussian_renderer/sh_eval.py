"""Real spherical-harmonics evaluation used by the ``convert_SHs_python`` branch of render()
(reference: gaussian_splatting/utils/sh_utils.py:57-118 ``eval_sh``; constants :24-55). Degrees 0..3 (the rasterizer's
own limit, cuda_rasterizer/forward.cu:22-73); coefficient layout [..., C, (deg+1)^2]."""
import torch

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[..., (deg+1)^2] basis values at unit directions [..., 3], in the reference's coefficient order and signs."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, _C0)]
    if deg > 0:
        b += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_C2[0] * xy, _C2[1] * yz, _C2[2] * (2.0 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if deg > 2:
        b += [_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
              _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy),
              _C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, -1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [..., C, K>=(deg+1)^2], dirs [..., 3] -> [..., C]."""
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise ValueError("not enough SH coefficients for this degree")
    return (sh[..., :n] * sh_basis(deg, dirs)[..., None, :]).sum(-1)
