"""oracle/torch_raster.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

An INDEPENDENT differentiable restatement of the reference rasterizer's forward in pure
PyTorch tensor ops (tile-binned, one dense [n_tile, 256] block per 16x16 tile), whose
backward comes from autograd rather than from the hand-derived formulas of
DGR/cuda_rasterizer/backward.cu. It has two jobs:

  1. cross-check the analytic backward of oracle/gs_oracle.c (and through it the HIP
     kernels) for means3D / scales / rotations / opacity / SH / colors_precomp / cov3D /
     means2D, and -- with ``viewmatrix`` differentiable through SE3_exp -- the pose
     gradient under the conditions where the reference's approximate pose Jacobians are
     exact;
  2. be the "host-CPU PyTorch fallback rasterizer" of BASELINE.json configs[0]
     (10k Gaussians @320x240, render + loss.backward() on CPU) and a CPU baseline.

Forward semantics follow DGR/cuda_rasterizer/forward.cu:157-392 and
rasterizer_impl.cu:198-344. Three places deliberately mirror
what the reference BACKWARD differentiates rather than what autograd would give for the
literal forward (each is a documented quirk of the reference):
  * Q4  the fov clamp of t.x/t.y: the clamped value is treated as a constant
        (backward.cu:182-183,269-271);
  * Q23 alpha = min(0.99, o*G) has no gradient mask (backward.cu:687-688,746-757):
        straight-through;
  * Q5  the conic backward uses 1/(det^2+1e-7) (backward.cu:210): not mirrored here, it
        is a <=1e-5 relative effect and stays inside the test tolerance.
"""
from __future__ import annotations

import math

import torch

BLOCK = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _sh_to_rgb(deg, shs, dirs):
    """forward.cu:22-73; shs [P,M,3], dirs [P,3] normalised."""
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
                   + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def _cov3d(scales, mod, rot):
    """forward.cu:120-154, quaternion not normalised."""
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    RS = R * (mod * scales)[:, None, :]
    return RS @ RS.transpose(1, 2)


def rasterize(means3D, means2D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
              sh_degree=0, scale_modifier=1.0, tile_window=None):
    """Differentiable forward. Returns (color[3,H,W], radii[P] i32, depth[1,H,W], opacity[1,H,W], n_touched[P] i32).

    ``means2D`` is the reference's dummy [P,3] zeros tensor whose gradient is d L / d(NDC mean).
    All tensors live on one device/dtype (cpu, fp32 or fp64). viewmatrix/projmatrix may require grad.
    ``tile_window`` = (tx0, ty0, tx1, ty1): composite only that rectangle of 16x16 tiles (everything else stays background) --
    bench.py's bounded CPU-baseline sample; the per-Gaussian preprocess always covers all P.
    """
    dev, dt = means3D.device, means3D.dtype
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    out_c = torch.zeros(3, H, W, device=dev, dtype=dt)
    out_d = torch.zeros(1, H, W, device=dev, dtype=dt)
    out_o = torch.zeros(1, H, W, device=dev, dtype=dt)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    n_touched = torch.zeros(P, dtype=torch.int32, device=dev)
    if P == 0:
        return out_c, radii, out_d, out_o, n_touched

    ones = torch.ones(P, 1, device=dev, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ viewmatrix[:, :3]                     # auxiliary.h:58-66 (row-vector convention)
    p_hom = hom @ projmatrix                             # auxiliary.h:68-77
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None] + means2D[:, :2]   # dummy means2D carries the NDC gradient
    in_front = p_view[:, 2] > 0.2                        # auxiliary.h:154

    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(-1, 3, 3)
    else:
        Sigma = _cov3d(scales, scale_modifier, rotations)

    # forward.cu:82-114
    tz = p_view[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], -1).reshape(-1, 2, 3)
    Rcw = viewmatrix[:3, :3].t()                         # R_cw[i][k] = vm[k][i]
    A = J @ Rcw
    cov2 = A @ Sigma @ A.transpose(1, 2)
    ca, cb, cc = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = ca * cc - cb * cb
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([cc / det_s, -cb / det_s, ca / det_s], -1)
    with torch.no_grad():
        mid = 0.5 * (ca + cc)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        rad = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2)))
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], -1)  # auxiliary.h:41-44
    with torch.no_grad():
        radi = torch.where(ok, rad, torch.zeros_like(rad)).to(torch.int64)
        pxd, pyd = pix[:, 0].detach(), pix[:, 1].detach()
        trunc = lambda v: torch.trunc(v).to(torch.int64)
        rminx = trunc((pxd - radi) / BLOCK).clamp(0, gx)
        rminy = trunc((pyd - radi) / BLOCK).clamp(0, gy)
        rmaxx = trunc((pxd + radi + BLOCK - 1) / BLOCK).clamp(0, gx)
        rmaxy = trunc((pyd + radi + BLOCK - 1) / BLOCK).clamp(0, gy)
        vis = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        radii = torch.where(vis, radi, torch.zeros_like(radi)).to(torch.int32)

    if colors_precomp is not None:
        feat = colors_precomp
    else:
        d = means3D - campos[None, :]
        feat = _sh_to_rgb(sh_degree, shs, d / d.norm(dim=1, keepdim=True))
    depth = p_view[:, 2]
    opac = opacities.reshape(-1)

    # global (depth, index) order == per-tile order of the stable (tile|depth) sort, rasterizer_impl.cu:98-108,306-311
    order = torch.argsort(depth.detach().to(torch.float32), stable=True)
    lx = torch.arange(BLOCK, device=dev)
    cols, deps, ops = [], [], []
    touched = torch.zeros(P, dtype=torch.int64, device=dev)
    color_rows = []
    for tyi in range(gy):
        row_c, row_d, row_o = [], [], []
        for txi in range(gx):
            if tile_window is not None and not (tile_window[0] <= txi < tile_window[2] and tile_window[1] <= tyi < tile_window[3]):
                ids = order[:0]
            else:
                m = vis & (rminx <= txi) & (txi < rmaxx) & (rminy <= tyi) & (tyi < rmaxy)
                ids = order[m[order]]
            pxs = (txi * BLOCK + lx).to(dt)
            pys = (tyi * BLOCK + lx).to(dt)
            PX, PY = torch.meshgrid(pxs, pys, indexing="xy")      # [16(y),16(x)]
            PX, PY = PX.reshape(-1), PY.reshape(-1)
            n = ids.numel()
            if n == 0:
                Tfin = torch.ones(BLOCK * BLOCK, device=dev, dtype=dt)
                Cc = torch.zeros(3, BLOCK * BLOCK, device=dev, dtype=dt)
                Dd = torch.zeros(BLOCK * BLOCK, device=dev, dtype=dt)
            else:
                dx = pix[ids, 0:1] - PX[None, :]
                dy = pix[ids, 1:2] - PY[None, :]
                co = conic[ids]
                power = -0.5 * (co[:, 0:1] * dx * dx + co[:, 2:3] * dy * dy) - co[:, 1:2] * dx * dy
                a_raw = opac[ids][:, None] * torch.exp(torch.clamp_max(power, 0.0))
                alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()     # Q23 straight-through
                valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
                a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                one_m = 1.0 - a_eff
                Tincl = torch.cumprod(one_m, 0)
                Texcl = torch.cat([torch.ones(1, Tincl.shape[1], device=dev, dtype=dt), Tincl[:-1]], 0)
                stop = (valid & (Tincl.detach() < 1e-4)).to(torch.int8).cummax(0).values.bool()  # forward.cu:357-362
                contrib = valid & ~stop
                w = torch.where(contrib, a_eff * Texcl, torch.zeros_like(a_eff))
                Cc = feat[ids].t() @ w
                Dd = (depth[ids][:, None] * w).sum(0)
                Tfin = torch.where(contrib, one_m, torch.ones_like(one_m)).prod(0)
                with torch.no_grad():
                    inside = ((PX < W) & (PY < H))[None, :]
                    cnt = (contrib & inside & (Tincl > 0.5)).sum(1)                    # forward.cu:369-371
                    touched.index_add_(0, ids, cnt)
            row_c.append((Cc + Tfin[None, :] * bg[:, None]).reshape(3, BLOCK, BLOCK))
            row_d.append(Dd.reshape(1, BLOCK, BLOCK))
            row_o.append((1.0 - Tfin).reshape(1, BLOCK, BLOCK))
        cols.append(torch.cat(row_c, 2)); deps.append(torch.cat(row_d, 2)); ops.append(torch.cat(row_o, 2))
    out_c = torch.cat(cols, 1)[:, :H, :W]
    out_d = torch.cat(deps, 1)[:, :H, :W]
    out_o = torch.cat(ops, 1)[:, :H, :W]
    return out_c, radii, out_d, out_o, touched.to(torch.int32)


def se3_exp(tau):
    """utils/pose_utils.py:28-77 restated (tau = [rho, theta])."""
    rho, th = tau[:3], tau[3:]
    zero = torch.zeros((), dtype=tau.dtype)
    Wm = torch.stack([zero, -th[2], th[1], th[2], zero, -th[0], -th[1], th[0], zero]).reshape(3, 3)
    W2 = Wm @ Wm
    ang = torch.sqrt((th * th).sum() + 1e-300)
    I = torch.eye(3, dtype=tau.dtype)
    if float(ang) < 1e-5:
        R = I + Wm + 0.5 * W2
        V = I + 0.5 * Wm + W2 / 6.0
    else:
        R = I + torch.sin(ang) / ang * Wm + (1 - torch.cos(ang)) / ang ** 2 * W2
        V = I + Wm * ((1 - torch.cos(ang)) / ang ** 2) + W2 * ((ang - torch.sin(ang)) / ang ** 3)
    T = torch.eye(4, dtype=tau.dtype)
    T = torch.cat([torch.cat([R, (V @ rho)[:, None]], 1), torch.tensor([[0, 0, 0, 1.0]], dtype=tau.dtype)], 0)
    return T
