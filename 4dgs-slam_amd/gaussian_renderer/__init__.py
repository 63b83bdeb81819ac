"""render() / render_flow() / get_dynamic_mask(): the Python wrapper between the SLAM code and the rasterizer -- a
counterpart of the reference's ``gaussian_splatting/gaussian_renderer/__init__.py`` (render :41-226, render_flow
:229-361, get_dynamic_mask :364-414) with the same signatures, the same dict keys and the same tensor semantics, so
``utils/slam_frontend.py``, ``utils/slam_backend.py`` and ``utils/eval_utils.py`` can call it unchanged. It imports the
MI355X rasterizer through the reference's own package name ``diff_gaussian_rasterization``.

``pc`` is duck-typed (the reference's GaussianModel): get_xyz, get_opacity, get_scaling, get_rotation, get_features,
active_sh_degree, max_sh_degree, dygs (bool[P] dynamic-Gaussian mask), and for the optional branches get_covariance(),
_deformation(...), _scaling, _rotation, _opacity, scaling_activation, rotation_activation.
``viewpoint_camera``: FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
projection_matrix, camera_center, cam_rot_delta, cam_trans_delta, time.
"""
import math
import os

import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

try:   # the MI355X package has the fused prologue; the CPU oracle stand-in used by the tests does not
    from diff_gaussian_rasterization import raw as _raw
except Exception:  # pragma: no cover
    _raw = None
try:   # ... and the multi-view entry point
    from diff_gaussian_rasterization import views as _views
except Exception:  # pragma: no cover
    _views = None

# render() hands the model's RAW parameters to the rasterizer when it can (diff_gaussian_rasterization/raw.py): the
# activations, the delta scatter and their autograd replay then run inside the kernels. GSR_FUSED_PROLOGUE=0 (or setting this
# flag to False) keeps the reference's chain of torch kernels.
FUSED_PROLOGUE = os.environ.get("GSR_FUSED_PROLOGUE", "1") != "0"

from .sh_eval import eval_sh


# ---- quaternion helpers exported by the reference module (:24-38) ------------------------------------------------
def standardize_quaternion(quaternions: torch.Tensor) -> torch.Tensor:
    """Flip the sign so the real part is non-negative."""
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product, real part first."""
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return standardize_quaternion(quaternion_raw_multiply(a, b))


# ---- shared pieces ------------------------------------------------------------------------------------------------
def _screenspace_points(pc):
    """Zero [P,3] tensor whose .grad receives dL/d(NDC mean) -- read by add_densification_stats (reference :69-78)."""
    pts = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=pc.get_xyz.device) + 0
    try:
        pts.retain_grad()
    except Exception:
        pass
    return pts


class _RenderPackage(dict):
    """render()'s dict whose "visibility_filter" (radii > 0) is formed when somebody reads it: the batched callers of the mapping back-end
    never do, and one comparison launch per view and per flow direction is ~25 launches of a dynamic mapping iteration."""

    def __missing__(self, key):
        if key != "visibility_filter":
            raise KeyError(key)
        value = self["radii"] > 0
        self[key] = value
        return value


_ZERO_BG = {}


def _zero_background(device):
    """The (0, 0, 0) background of render_flow, one tensor per device (never written)."""
    key = (device.type, device.index)
    if key not in _ZERO_BG:
        _ZERO_BG[key] = torch.zeros(3, device=device)
    return _ZERO_BG[key]


def _settings(cam, bg, scaling_modifier, sh_degree):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=bg, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, projmatrix_raw=cam.projection_matrix,
        sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)


def _scatter_delta(like: torch.Tensor, index: torch.Tensor, delta) -> torch.Tensor:
    """zeros_like(like) with `delta` written into the rows selected by the boolean mask `index` (reference :163-174)."""
    full = torch.zeros_like(like)
    full[index] = delta
    return full


def _python_colors(pc, cam):
    """SH -> RGB on the Python side (pipe.convert_SHs_python, reference :134-143)."""
    shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dir_pp = pc.get_xyz - cam.camera_center.repeat(pc.get_features.shape[0], 1)
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)


def _fused_prologue_ok(pc, pipe, mask, dynamic) -> bool:
    """The fused route reproduces exactly the branch of render() taken with the shipped configs: SH colours and covariance
    from scale/rotation in the rasterizer (pipe.convert_SHs_python = compute_cov3D_python = False, base_config.yaml), no
    and a GaussianModel whose activations are the reference's (scene/gaussian_model.py:60-68: exp, sigmoid, F.normalize).  With
    `dynamic` the 4DGaussians deformation network runs first and ITS outputs (deformed means, log-scales, raw rotations) take the
    place of the raw parameters; the combination dynamic + control-node deltas keeps the reference's torch chain (:159-174 then
    overwrite part of the network's result)."""
    if not FUSED_PROLOGUE or _raw is None:
        return False
    if getattr(pipe, "compute_cov3D_python", False) or getattr(pipe, "convert_SHs_python", False):
        return False
    try:
        raws = (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest)
        acts = (pc.scaling_activation is torch.exp and pc.opacity_activation is torch.sigmoid
                and pc.rotation_activation is torch.nn.functional.normalize)
    except AttributeError:
        return False
    return acts and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in raws)


def _deform(pc, viewpoint_camera, means3D, shs):
    """The 4DGaussians deformation field (:149-157): (deformed means, log-scales, raw rotations)."""
    time = torch.tensor(viewpoint_camera.time).to(means3D.device).repeat(means3D.shape[0], 1)
    raw_scaling = pc._scaling.repeat(1, 3) if pc._scaling.shape[-1] == 1 else pc._scaling
    means3D, scales_final, rotations_final, _, _, _ = pc._deformation(means3D, raw_scaling, pc._rotation, pc._opacity, shs, time)
    return means3D, scales_final, rotations_final


def _mask_rows(mask):
    """int32 row list of render()'s boolean `mask`. A caller that renders with the same mask many times (the tracking loop) can attach
    the list once as ``mask._gsr_gather`` and save the nonzero() -- a host synchronisation -- per call."""
    g = getattr(mask, "_gsr_gather", None)
    return g if g is not None else _raw.gather_from_mask(mask)


def _render_fused(viewpoint_camera, pc, bg_color, scaling_modifier, screenspace_points, dx, ds, dr, mask=None, dynamic=False):
    deltas = dx is not None and ds is not None and dr is not None        # the reference applies them only together (:159)
    slot = _dyn_slot(pc) if deltas else None
    f_rest = pc._features_rest if pc._features_rest.shape[1] > 0 else None
    xyz, log_scales, raw_rot = pc._xyz, pc._scaling, pc._rotation
    if dynamic:
        # the network's outputs are "raw parameters" too: exp / normalize (:155-156) and their chain rules happen in the kernels.
        # shs is handed over only if the network can use it (no_dshs False); the reference concatenates it for every call (:145)
        net_args = getattr(getattr(pc._deformation, "deformation_net", None), "args", None)
        shs = None if getattr(net_args, "no_dshs", False) else pc.get_features
        xyz, log_scales, raw_rot = _deform(pc, viewpoint_camera, pc.get_xyz, shs)
    return _raw.rasterize_gaussians_raw(
        _settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree), xyz, screenspace_points, log_scales,
        raw_rot, pc._opacity, pc._features_dc, f_rest, slot, dx if deltas else None, ds if deltas else None,
        dr if deltas else None, viewpoint_camera.cam_rot_delta, viewpoint_camera.cam_trans_delta,
        gather=None if mask is None else _mask_rows(mask))


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, mask=None,
           dynamic=False, dx=None, ds=None, dr=None, do=None, dc=None, novel=0):
    """Render the scene. Returns None for an empty model, else the dict
    {render, viewspace_points, visibility_filter, radii, depth, opacity, n_touched} (reference :218-226)."""
    if pc.get_xyz.shape[0] == 0:
        return None
    screenspace_points = _screenspace_points(pc)
    deltas = dx is not None and ds is not None and dr is not None
    if deltas and all(isinstance(v, (int, float)) and v == 0 for v in (dx, ds, dr)):
        # the evaluation call shape `dx = ds = dr = 0` (utils/eval_utils.py:339-344): the reference scatters zeros and adds them,
        # which changes nothing -- same as no deltas
        dx = ds = dr = None
        deltas = False
    tensor_deltas = not deltas or all(isinstance(v, torch.Tensor) for v in (dx, ds, dr))
    if _fused_prologue_ok(pc, pipe, mask, dynamic) and not (dynamic and deltas) and tensor_deltas:
        rendered_image, radii, depth, opacity, n_touched = _render_fused(viewpoint_camera, pc, bg_color, scaling_modifier,
                                                                         screenspace_points, dx, ds, dr, mask, dynamic)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
                "depth": depth, "opacity": opacity, "n_touched": n_touched}
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree))

    means3D, means2D, opacity = pc.get_xyz, screenspace_points, pc.get_opacity

    # covariance: Python-side precompute, or scales (+ isotropic expansion) / rotations for the kernel (:116-127)
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling.repeat(1, 3) if pc.get_scaling.shape[-1] == 1 else pc.get_scaling
        rotations = pc.get_rotation

    # colour: the reference never forwards override_color (its `colors_precomp is None` test is always true, :131-147)
    shs = colors_precomp = None
    if pipe.convert_SHs_python:
        colors_precomp = _python_colors(pc, viewpoint_camera)
    else:
        shs = pc.get_features

    if dynamic:  # 4DGaussians deformation field (:149-157)
        means3D, scales_final, rotations_final = _deform(pc, viewpoint_camera, means3D, shs)
        scales = pc.scaling_activation(scales_final)
        rotations = pc.rotation_activation(rotations_final)

    if dx is not None and ds is not None and dr is not None:  # control-node deltas on the dynamic subset (:159-174)
        means3D = pc.get_xyz + _scatter_delta(means3D, pc.dygs, dx)
        scales = scales + _scatter_delta(scales, pc.dygs, ds)
        rotations = pc.get_rotation + _scatter_delta(rotations, pc.dygs, dr)   # not re-normalised

    sel = (lambda t: t[mask]) if mask is not None else (lambda t: t)
    opt = lambda t: None if t is None else sel(t)
    kwargs = dict(
        means3D=sel(means3D), means2D=sel(means2D),
        shs=sel(shs) if mask is not None else shs,          # with a mask the reference indexes shs unconditionally (:183)
        colors_precomp=opt(colors_precomp), opacities=sel(opacity),
        scales=sel(scales) if mask is not None else scales,  # likewise scales / rotations (:186-187)
        rotations=sel(rotations) if mask is not None else rotations,
        cov3D_precomp=opt(cov3D_precomp),
        theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta)
    rendered_image, radii, depth, opacity, n_touched = rasterizer(**kwargs)
    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "depth": depth,
        "opacity": opacity,
        "n_touched": n_touched,
    }


def _render_views_dynamic(cams, pc, pipe, bg_color, scaling_modifier):
    """render(dynamic=True) of the cameras of one mapping iteration at once, or None when the batched route does not apply: the 4DGaussians
    deformation network is evaluated ONCE for all cameras' times (deform_network.forward_views: the position-only planes of the HexPlane
    field gathered once per Gaussian, one sort and one spatial scatter on the way back, the MLP over all V * P rows), and its [V, P, 10]
    output goes to the multi-view rasterizer entry point as per-view deltas in front of the activations (views.rasterize_views_net)."""
    if (_views is None or os.environ.get("GSR_MULTI_VIEW", "1") == "0" or os.environ.get("GSR_DYNAMIC_VIEWS", "1") == "0" or len(cams) < 2
            or len(cams) > _views.MAX_VIEWS or not _fused_prologue_ok(pc, pipe, None, True)):
        return None
    net = getattr(pc, "_deformation", None)
    if net is None or not hasattr(net, "forward_views"):
        return None
    settings = [_settings(c, bg_color, scaling_modifier, pc.active_sh_degree) for c in cams]
    if not _views.views_supported(settings):
        return None
    net_out = net.forward_views(pc.get_xyz, [float(c.time) for c in cams])
    if net_out is None:
        return None
    block = torch.zeros((len(cams),) + tuple(pc.get_xyz.shape), dtype=pc.get_xyz.dtype, device=pc.get_xyz.device)
    points = [block[v].requires_grad_(True) for v in range(len(cams))]
    f_rest = pc._features_rest if pc._features_rest.shape[1] > 0 else None
    outs = _views.rasterize_views_net(settings, pc._xyz, points, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, f_rest, net_out,
                                      poses=[(c.cam_rot_delta, c.cam_trans_delta) for c in cams])
    return [_RenderPackage({"render": o[0], "viewspace_points": pts, "radii": o[1], "depth": o[2], "opacity": o[3], "n_touched": o[4]})
            for o, pts in zip(outs, points)]


def render_views(viewpoint_cameras, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, deltas=None, dynamic=False):
    """render() of several cameras of one mapping iteration at once: a list of render()'s dicts, one per camera.
    ``dynamic``: render(dynamic=True) -- every camera's Gaussians moved by the deformation network at the camera's time (not combined with
    ``deltas``, like the fused single-camera route); batched through _render_views_dynamic when that applies, else one render() per camera.

    The mapping back-end renders the same Gaussians from every window keyframe and two random ones before each optimizer step
    (utils/slam_backend.py:357,526,657); with the fused prologue available those views go through the multi-view entry point
    (diff_gaussian_rasterization/views.py: one launch per pipeline stage for all of them), otherwise -- Python-side SH / covariance,
    a single camera, more than views.MAX_VIEWS, cameras of different size -- one render() call per camera. ``deltas``: per camera
    None or the (dx, ds, dr) tensors of the dynamic subset; values and gradients are those of the per-camera calls."""
    cams = list(viewpoint_cameras)
    if dynamic:
        if deltas is not None and any(d is not None for d in deltas):
            return [render(c, pc, pipe, bg_color, scaling_modifier, dynamic=True, dx=d[0] if d else None, ds=d[1] if d else None, dr=d[2] if d else None)
                    for c, d in zip(cams, deltas)]
        batched = _render_views_dynamic(cams, pc, pipe, bg_color, scaling_modifier) if pc.get_xyz.shape[0] else None
        return batched if batched is not None else [render(c, pc, pipe, bg_color, scaling_modifier, dynamic=True) for c in cams]
    deltas = list(deltas) if deltas is not None else [None] * len(cams)
    per_camera = lambda: [render(c, pc, pipe, bg_color, scaling_modifier, dx=d[0] if d else None, ds=d[1] if d else None, dr=d[2] if d else None)
                          for c, d in zip(cams, deltas)]
    if pc.get_xyz.shape[0] == 0 or len(cams) < 2 or _views is None or os.environ.get("GSR_MULTI_VIEW", "1") == "0":
        return per_camera()
    if not _fused_prologue_ok(pc, pipe, None, False):
        return per_camera()
    for d in deltas:
        if d is not None and not (len(d) == 3 and all(isinstance(v, torch.Tensor) for v in d)):
            return per_camera()
    settings = [_settings(c, bg_color, scaling_modifier, pc.active_sh_degree) for c in cams]
    if not _views.views_supported(settings):
        return per_camera()
    block = torch.zeros((len(cams),) + tuple(pc.get_xyz.shape), dtype=pc.get_xyz.dtype, device=pc.get_xyz.device)    # one fill for all views
    points = [block[v].requires_grad_(True) for v in range(len(cams))]     # leaves: .grad receives the view's dL/d(NDC mean) (reference :69-78)
    any_delta = any(d is not None for d in deltas)
    f_rest = pc._features_rest if pc._features_rest.shape[1] > 0 else None
    outs = _views.rasterize_views_raw(settings, pc._xyz, points, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, f_rest,
                                      dyn_slot=_dyn_slot(pc) if any_delta else None, deltas=deltas,
                                      poses=[(c.cam_rot_delta, c.cam_trans_delta) for c in cams])
    return [_RenderPackage({"render": o[0], "viewspace_points": pts, "radii": o[1], "depth": o[2], "opacity": o[3], "n_touched": o[4]})
            for o, pts in zip(outs, points)]


def _flow_fused_ok(pc) -> bool:
    if not FUSED_PROLOGUE or _raw is None or os.environ.get("GSR_FUSED_FLOW", "1") == "0":
        return False
    try:
        raws = (pc._xyz, pc._scaling, pc._rotation, pc._opacity)
        acts = (pc.scaling_activation is torch.exp and pc.opacity_activation is torch.sigmoid and pc.rotation_activation is torch.nn.functional.normalize)
    except AttributeError:
        return False
    # render_flow has no isotropic expansion in the reference (its [P,1] scaling cannot take a [K,3] d_scaling): anisotropic models only
    return acts and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in raws) and pc._scaling.shape[-1] == 3


def _dyn_slot(pc):
    """int32 slot map of pc.dygs, cached on the mask tensor (it only changes when the model is rebuilt or the mask is written)."""
    m = pc.dygs
    hit = getattr(m, "_gsr_slot", None)
    if hit is not None and hit[0] == m._version:
        return hit[1]
    s = _raw.dyn_slot_from_mask(m)
    try:
        m._gsr_slot = (m._version, s)
    except Exception:
        pass
    return s


def render_flow_views(pc, requests, scaling_modifier=1.0, clips=None):
    """render_flow for several (camera 1, camera 2) pairs of one mapping iteration: requests[i] = (viewpoint_camera1, viewpoint_camera2,
    d_xyz1, d_xyz2, d_rotation1, d_scaling1), the arguments of render_flow in its order; returns render_flow's dict per request. The
    dynamic mapping loop renders two flow images per window keyframe before each optimizer step (utils/slam_backend.py:486,496); with
    the fused route available they go through the multi-view entry point in groups of up to views.MAX_VIEWS (one launch per pipeline
    stage per group), otherwise one render_flow call each. ``clips``: per request None or an int32 [4] device tensor, the tile rectangle
    [x0, y0, x1, y1) the caller will read of that image (its loss mask's bounding box): Gaussians outside it are culled on the batched route
    (gsr_view.flow_clip -- an argument of the call, nothing is left behind if it raises; pixels inside the rectangle and the gradients of a loss
    confined to them are unchanged)."""
    requests = list(requests)
    single = lambda: [render_flow(pc, c1, c2, dx1, dx2, dr1, ds1, scaling_modifier=scaling_modifier) for c1, c2, dx1, dx2, dr1, ds1 in requests]
    if (len(requests) < 2 or _views is None or os.environ.get("GSR_MULTI_VIEW", "1") == "0" or not _flow_fused_ok(pc) or pc.get_xyz.shape[0] == 0
            or not all(isinstance(t, torch.Tensor) for r in requests for t in r[2:])):
        return single()
    bg = _zero_background(pc.get_xyz.device)
    settings = [_settings(r[0], bg, scaling_modifier, 0) for r in requests]
    slot = _dyn_slot(pc)
    out = []
    for lo in range(0, len(requests), _views.MAX_VIEWS):
        part, rs = requests[lo:lo + _views.MAX_VIEWS], settings[lo:lo + _views.MAX_VIEWS]
        # (the chunks of one iteration keep separate capacity estimates per view: include/gs_rasterizer.h "view_slot_group")
        group_before = _views._C.set_option("view_slot_group", lo // _views.MAX_VIEWS)
        try:
            out += _render_flow_chunk(pc, part, rs, slot, scaling_modifier, None if clips is None else list(clips[lo:lo + _views.MAX_VIEWS]))
        finally:
            _views._C.set_option("view_slot_group", group_before)
    return out


def _render_flow_chunk(pc, part, rs, slot, scaling_modifier, clips=None):
    if len(part) == 1 or not _views.views_supported(rs):           # (cameras of different size / field of view: one call each)
        return [render_flow(pc, c1, c2, dx1, dx2, dr1, ds1, scaling_modifier=scaling_modifier) for c1, c2, dx1, dx2, dr1, ds1 in part]
    block = torch.zeros((len(part),) + tuple(pc.get_xyz.shape), dtype=pc.get_xyz.dtype, device=pc.get_xyz.device)
    points = [block[v].requires_grad_(True) for v in range(len(part))]
    flows = [(dx1, dx2, ds1, dr1, c1.full_proj_transform, (c2 if c2 is not None else c1).full_proj_transform) for c1, c2, dx1, dx2, dr1, ds1 in part]
    res = _views.rasterize_flow_views_raw(rs, pc._xyz, points, pc._scaling.detach(), pc._rotation.detach(), pc._opacity.detach(), slot, flows, clips=clips)
    return [_RenderPackage({"render": o[0], "depth": o[2], "alpha": o[3], "viewspace_points": pts, "radii": o[1]}) for o, pts in zip(res, points)]


def render_flow(pc, viewpoint_camera1, viewpoint_camera2, d_xyz1, d_xyz2, d_rotation1, d_scaling1, scaling_modifier=1.0,
                compute_cov3D_python=False, scale_const=None, d_rot_as_res=True, **kwargs):
    """Rasterize (NDC flow u, NDC flow v, dynamic mask) as colours (reference :229-361). Flow is computed from DETACHED
    canonical positions plus the attached deltas (:262); the rasterized means stay attached to pc.get_xyz (:261,305)."""
    cam2 = viewpoint_camera2 if viewpoint_camera2 is not None else viewpoint_camera1
    fused = (_flow_fused_ok(pc) and scale_const is None and not compute_cov3D_python and d_rot_as_res
             and all(isinstance(t, torch.Tensor) for t in (d_xyz1, d_xyz2, d_rotation1, d_scaling1)))
    # (fused: a plain zero leaf -- the reference's `zeros_like(..) + 0` with retain_grad is one more launch each way for the same .grad)
    screenspace_points = torch.zeros_like(pc.get_xyz, requires_grad=True) if fused else _screenspace_points(pc)
    if fused:
        # fused route (diff_gaussian_rasterization/raw.py rasterize_flow_raw): both projections, the NDC flow, the mask channel and all
        # scatter-adds happen inside preprocess_fwd / geometry_bwd -- ~15 elementwise torch kernels and 4 index_puts per call less,
        # and the dynamic mapping loop calls this twice per view (utils/slam_backend.py:486,496)
        rs = _settings(viewpoint_camera1, _zero_background(pc.get_xyz.device), scaling_modifier, 0)
        slot = _dyn_slot(pc)
        rendered_image, radii, rendered_depth, rendered_alpha, n_touched = _raw.rasterize_flow_raw(
            rs, pc._xyz, screenspace_points, pc._scaling.detach(), pc._rotation.detach(), pc._opacity.detach(), slot, d_xyz1, d_xyz2, d_scaling1,
            d_rotation1, viewpoint_camera1.full_proj_transform, cam2.full_proj_transform)
        return _RenderPackage({"render": rendered_image, "depth": rendered_depth, "alpha": rendered_alpha, "viewspace_points": screenspace_points,
                               "radii": radii})
    canonical_xyz = pc.get_xyz.clone()
    base = canonical_xyz.detach()
    dxyz1 = _scatter_delta(base, pc.dygs, d_xyz1)
    dxyz2 = _scatter_delta(base, pc.dygs, d_xyz2)
    xyz_t1, xyz_t2 = base + dxyz1, base + dxyz2

    def project(xyz, full_proj):
        hom = torch.cat([xyz, torch.ones_like(xyz[..., :1])], dim=-1) @ full_proj
        return hom[..., :3] / (hom[..., -1:] + 1e-7)

    proj2 = viewpoint_camera2.full_proj_transform if viewpoint_camera2 is not None else viewpoint_camera1.full_proj_transform
    flow_uvz = project(xyz_t2, proj2) - project(xyz_t1, viewpoint_camera1.full_proj_transform)
    flow_uvz[..., -1:] = pc.dygs.unsqueeze(1)   # third channel renders the motion mask (:284)

    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera1, torch.zeros_like(flow_uvz[0]), scaling_modifier, 0))
    means3D = canonical_xyz + dxyz1
    opacity = pc.get_opacity.clone().detach()

    scales = rotations = cov3D_precomp = None
    residual_rot = lambda: pc.get_rotation if type(d_rotation1) is float else quaternion_multiply(d_rotation1, pc.get_rotation)
    if scale_const is not None:
        scales = torch.ones_like(pc.get_scaling) * scale_const
        rotations = pc.get_rotation + d_rotation1 if d_rot_as_res else residual_rot()
    elif compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier, d_rotation=None if type(d_rotation1) is float else d_rotation1)
    else:
        scales = pc.get_scaling.clone().detach() + _scatter_delta(pc.get_scaling, pc.dygs, d_scaling1)
        if d_rot_as_res:
            rotations = pc.get_rotation.clone().detach() + _scatter_delta(pc.get_rotation, pc.dygs, d_rotation1)
        else:
            rotations = residual_rot()

    rendered_image, radii, rendered_depth, rendered_alpha, n_touched = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=None, colors_precomp=flow_uvz, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {
        "render": rendered_image,
        "depth": rendered_depth,
        "alpha": rendered_alpha,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
    }


def get_dynamic_mask(viewpoint_camera, pc, pipe, override_color=None, dynamic=True):
    """bool[P] static mask from the deformation field's displacement magnitudes (reference :364-414)."""
    if pc.get_xyz.shape[0] == 0:
        return None
    means3D = pc.get_xyz.clone().detach()
    time = torch.tensor(viewpoint_camera.time - 1).to(means3D.device).repeat(means3D.shape[0], 1)
    shs = None if pipe.convert_SHs_python else pc.get_features
    if not dynamic:
        return None
    if pc.get_scaling.shape[-1] == 1:
        _, _, _, dx, ds, dr = pc._deformation(means3D, pc._scaling.repeat(1, 3).clone().detach(), pc._rotation.clone().detach(),
                                              pc._opacity.clone().detach(), shs.clone().detach(), time)
    else:
        _, _, _, dx, ds, dr = pc._deformation(means3D, pc._scaling, pc._rotation, pc._opacity, shs, time)
    return (torch.norm(dx, dim=1) < 1) & (torch.norm(ds, dim=1) < 2) & (torch.norm(dr, dim=1) < 1)
