"""ctypes declarations of include/slam_map.h (seeding, densification, camera step) on libgs_rasterizer_hip.so. No CPU path."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C

_vp, _f, _i = C.c_void_p, C.c_float, C.c_int


class DensifyTensor(C.Structure):       # gsr_densify_tensor
    _fields_ = [("src", _vp), ("dst", _vp), ("width", _i), ("kind", _i)]


class CameraStep(C.Structure):          # gsr_camera_step
    _fields_ = [("rot_delta", _vp), ("g_rot_delta", _vp), ("trans_delta", _vp), ("g_trans_delta", _vp),
                ("exposure_a", _vp), ("g_exposure_a", _vp), ("exposure_b", _vp), ("g_exposure_b", _vp),
                ("exp_avg", _vp), ("exp_avg_sq", _vp), ("step", _vp),
                ("lr_rot", _f), ("lr_trans", _f), ("lr_exposure", _f), ("beta1", _f), ("beta2", _f), ("eps", _f),
                ("R", _vp), ("T", _vp), ("projmatrix", _vp), ("viewmatrix", _vp), ("full_proj", _vp), ("campos", _vp),
                ("converged", _vp), ("converged_threshold", _f), ("do_pose", _i), ("latch", _i)]


class TrackLoss(C.Structure):           # gsr_track_loss
    _fields_ = [("gt_image", _vp), ("gt_depth", _vp), ("w_rgb", _vp), ("w_depth", _vp), ("alpha", _f), ("opacity_depth_threshold", _f),
                ("opacity_weights", _i)]


class KeyframeEntry(C.Structure):       # gsr_keyframe_entry
    _fields_ = [(n, _vp) for n in ("viewmatrix", "full_proj", "campos", "exposure_a", "exposure_b", "gt_image", "gt_depth", "w_rgb", "w_depth")]


CAMERA_STEPS_MAX, SLOTS_MAX = 12, 4
COPY, STATE, XYZ, SCALE = 0, 1, 2, 3
_declared = False


def lib():
    global _declared
    L = _C.load_library()
    if not _declared:
        L.gsr_seed_workspace_size.restype = C.c_size_t
        L.gsr_seed_workspace_size.argtypes = [_i]
        L.gsr_seed_from_rgbd.restype = _i
        L.gsr_seed_from_rgbd.argtypes = [_i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.gsr_densify_select.restype = _i
        L.gsr_densify_select.argtypes = [_i, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _vp, _vp]
        L.gsr_densify_apply.restype = _i
        L.gsr_densify_apply.argtypes = [_i, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(DensifyTensor), _vp, _vp, _i, _vp, _vp, _vp]
        L.gsr_camera_step_launch.restype = _i
        L.gsr_camera_step_launch.argtypes = [C.POINTER(CameraStep), _vp]
        L.gsr_camera_steps_launch.restype = _i
        L.gsr_camera_steps_launch.argtypes = [_i, C.POINTER(CameraStep), _vp]
        from diff_gaussian_rasterization.raw import _RawInputs
        L.gsr_track_workspace_size.restype = C.c_size_t
        L.gsr_track_workspace_size.argtypes = [_i, _i]
        L.gsr_track_step.restype = _i
        L.gsr_track_step.argtypes = [_C._ALLOC_FN, _vp, _C._ALLOC_FN, _vp, _C._ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, C.POINTER(_RawInputs), _f, _vp,
                                     _f, _f, _vp, _vp, _vp, _vp, _vp, C.POINTER(TrackLoss), C.POINTER(CameraStep), _vp, _vp, _vp]
        L.gsr_schedule_advance.restype = _i
        L.gsr_schedule_advance.argtypes = [_vp, _vp, _i, _i, _vp, _vp]
        L.gsr_slot_gather.restype = _i
        L.gsr_slot_gather.argtypes = [_i, _vp, _vp, C.POINTER(KeyframeEntry), _i, _vp]
        L.gsr_isotropic_loss_workspace_size.restype = C.c_size_t
        L.gsr_isotropic_loss_workspace_size.argtypes = [_i]
        L.gsr_isotropic_loss_forward.restype = _i
        L.gsr_isotropic_loss_forward.argtypes = [_i, _vp, _vp, _vp, _vp]
        L.gsr_isotropic_loss_backward.restype = _i
        L.gsr_isotropic_loss_backward.argtypes = [_i, _vp, _vp, _vp, _vp]
        L.gsr_arap_forward.restype = _i
        L.gsr_arap_forward.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
        L.gsr_arap_backward.restype = _i
        L.gsr_arap_backward.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.gsr_elastic_forward.restype = _i
        L.gsr_elastic_forward.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp]
        L.gsr_elastic_backward.restype = _i
        L.gsr_elastic_backward.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
        L.gsr_kabsch_rotations.restype = _i
        L.gsr_kabsch_rotations.argtypes = [_i, _vp, _vp, _vp]
        L.gsr_edge_mask.restype = _i
        L.gsr_edge_mask.argtypes = [_vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp]
        _declared = True
    return L


def check(rc, what):
    if rc < 0:
        _C._err(lib(), rc, what)


def stream(dev):
    return _C._stream(dev)


def dev_f32(t, name):
    _C._require_device(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous float32 device tensor")
    return t.data_ptr()
