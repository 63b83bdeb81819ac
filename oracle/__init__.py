"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loader for ``oracle/libgs_oracle.so`` (built by ``oracle/Makefile`` from
``gs_oracle.c`` / ``knn_oracle.c``): the sequential CPU restatement of the reference's
CUDA rasterizer (submodules/diff-gaussian-rasterization) and of simple_knn.distCUDA2.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the CPU baseline. Nothing under
``4dgs-slam_amd/`` imports it.

Parity status: unpinned by reference outputs (the reference path is CUDA-only and ships
no golden vectors); see the header of gs_oracle.c for what pins it instead.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("gs_oracle.c", "knn_oracle.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_variant = "serial"


def set_variant(which: str = "serial", threads: int | None = None):
    """"serial" (the checker, default) or "omp": libgs_oracle_omp.so, the same fp32 restatement with its tile and per-Gaussian
    loops under OpenMP and atomic accumulation -- ONLY for bench.py's all-core cpu_baseline timing (summation order not fixed)."""
    global _lib, _variant
    assert which in ("serial", "omp")
    if which != _variant:
        _lib, _variant = None, which
    if which == "omp" and threads:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH if _variant == "serial" else os.path.join(_HERE, "libgs_oracle_omp.so"))
        for sfx in (("f32", "f64") if _variant == "serial" else ("f32",)):
            getattr(_lib, f"gso_forward_{sfx}").restype = C.c_void_p
            getattr(_lib, f"gso_num_rendered_{sfx}").restype = C.c_int
            getattr(_lib, f"gso_num_rendered_{sfx}").argtypes = [C.c_void_p]
            getattr(_lib, f"gso_free_{sfx}").argtypes = [C.c_void_p]
            getattr(_lib, f"gso_free_{sfx}").restype = None
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(x, dt, empty_is_none=True):
    """contiguous numpy array of dtype dt; None / empty -> None (the reference's nullptr convention)."""
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(x), dtype=dt)
    if empty_is_none and a.size == 0:
        return None
    return a


class OracleState:
    """Owns the C-side GeometryState/ImageState/BinningState of one forward call."""

    def __init__(self, handle, sfx, dt, P, M, W, H, args):
        self.handle, self.sfx, self.dt = handle, sfx, dt
        self.P, self.M, self.W, self.H = P, M, W, H
        self.args = args  # numpy inputs kept alive for backward
        self.num_rendered = getattr(lib(), f"gso_num_rendered_{sfx}")(handle)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                getattr(lib(), f"gso_free_{self.sfx}")(self.handle)
                self.handle = None
        except Exception:  # interpreter shutdown
            pass

    def state(self) -> dict:
        P, W, H, R, dt = self.P, self.W, self.H, self.num_rendered, self.dt
        T = ((W + 15) // 16) * ((H + 15) // 16)
        out = dict(
            depths=np.zeros(P, dt), clamped=np.zeros((P, 3), np.uint8), means2D=np.zeros((P, 2), dt),
            cov3D=np.zeros((P, 6), dt), conic_opacity=np.zeros((P, 4), dt), rgb=np.zeros((P, 3), dt),
            tiles_touched=np.zeros(P, np.uint32), point_offsets=np.zeros(P, np.uint32),
            final_T=np.zeros((H, W), dt), n_contrib=np.zeros((H, W), np.uint32),
            ranges=np.zeros((T, 2), np.uint32), keys=np.zeros(R, np.uint64), point_list=np.zeros(R, np.uint32),
        )
        order = ["depths", "clamped", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched",
                 "point_offsets", "final_T", "n_contrib", "ranges", "keys", "point_list"]
        fn = getattr(lib(), f"gso_get_state_{self.sfx}")
        fn.restype = None
        fn(C.c_void_p(self.handle), *[_ptr(out[k]) for k in order])
        return out


def rasterize_forward(*, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                      image_height, image_width, shs=None, colors_precomp=None, scales=None, rotations=None,
                      cov3D_precomp=None, scale_modifier=1.0, sh_degree=0, prefiltered=False,
                      dtype=np.float32):
    """Reference forward (rasterize_points.cu:35-122 + rasterizer_impl.cu:198-344) on the CPU.

    Returns (out dict, OracleState). Output shapes follow rasterize_points.cu:69-73.
    """
    dt = np.dtype(dtype)
    sfx = "f32" if dt == np.float32 else "f64"
    real = C.c_float if sfx == "f32" else C.c_double
    means3D = _arr(means3D, dt, empty_is_none=False)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    shs_a = _arr(shs, dt)
    M = 0 if shs_a is None else shs_a.shape[1]  # rasterize_points.cu:87-91
    a = dict(
        bg=_arr(bg, dt), means3D=means3D, shs=shs_a, colors=_arr(colors_precomp, dt), opac=_arr(opacities, dt),
        scales=_arr(scales, dt), rots=_arr(rotations, dt), cov=_arr(cov3D_precomp, dt), vm=_arr(viewmatrix, dt),
        pm=_arr(projmatrix, dt), cam=_arr(campos, dt), M=M, D=int(sh_degree), mod=float(scale_modifier),
        tfx=float(tanfovx), tfy=float(tanfovy),
    )
    out = dict(
        color=np.zeros((3, H, W), dt), depth=np.zeros((1, H, W), dt), opacity=np.zeros((1, H, W), dt),
        radii=np.zeros(P, np.int32), n_touched=np.zeros(P, np.int32),
    )
    if P == 0:  # rasterize_points.cu:85 -- outputs stay zero
        out["num_rendered"] = 0
        return out, None
    fn = getattr(lib(), f"gso_forward_{sfx}")
    h = fn(C.c_int(P), C.c_int(a["D"]), C.c_int(M), _ptr(a["bg"]), C.c_int(W), C.c_int(H),
           _ptr(a["means3D"]), _ptr(a["shs"]), _ptr(a["colors"]), _ptr(a["opac"]),
           _ptr(a["scales"]), real(a["mod"]), _ptr(a["rots"]), _ptr(a["cov"]),
           _ptr(a["vm"]), _ptr(a["pm"]), _ptr(a["cam"]), real(a["tfx"]), real(a["tfy"]), C.c_int(int(prefiltered)),
           _ptr(out["color"]), _ptr(out["depth"]), _ptr(out["opacity"]), _ptr(out["radii"]), _ptr(out["n_touched"]))
    if not h:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    st = OracleState(h, sfx, dt, P, M, W, H, a)
    out["num_rendered"] = st.num_rendered
    return out, st


def rasterize_backward(st: Optional[OracleState], *, projmatrix_raw, dL_dcolor, dL_ddepth, P=None, M=0,
                       dtype=np.float32):
    """Reference backward (rasterize_points.cu:124-211 + rasterizer_impl.cu:348-455) on the CPU."""
    if st is None:
        dt = np.dtype(dtype)
        P = P or 0
    else:
        dt, P, M = st.dt, st.P, st.M
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), dt), dL_dconic=np.zeros((P, 2, 2), dt), dL_dopacity=np.zeros((P, 1), dt),
        dL_dcolors=np.zeros((P, 3), dt), dL_ddepths=np.zeros((P, 1), dt), dL_dmeans3D=np.zeros((P, 3), dt),
        dL_dcov3D=np.zeros((P, 6), dt), dL_dsh=np.zeros((P, M, 3), dt), dL_dscales=np.zeros((P, 3), dt),
        dL_drotations=np.zeros((P, 4), dt), dL_dtau=np.zeros((P, 6), dt),
    )
    if st is None or P == 0:
        return g
    a = st.args
    real = C.c_float if st.sfx == "f32" else C.c_double
    dpix = _arr(dL_dcolor, dt, empty_is_none=False)
    ddep = _arr(dL_ddepth, dt, empty_is_none=False)
    praw = _arr(projmatrix_raw, dt)
    fn = getattr(lib(), f"gso_backward_{st.sfx}")
    fn.restype = None
    fn(C.c_void_p(st.handle), _ptr(a["bg"]), _ptr(a["means3D"]), _ptr(a["shs"]), _ptr(a["colors"]),
       _ptr(a["scales"]), real(a["mod"]), _ptr(a["rots"]), _ptr(a["cov"]),
       _ptr(a["vm"]), _ptr(a["pm"]), _ptr(praw), _ptr(a["cam"]), real(a["tfx"]), real(a["tfy"]),
       _ptr(dpix), _ptr(ddep),
       _ptr(g["dL_dmeans2D"]), _ptr(g["dL_dconic"]), _ptr(g["dL_dopacity"]), _ptr(g["dL_dcolors"]), _ptr(g["dL_ddepths"]),
       _ptr(g["dL_dmeans3D"]), _ptr(g["dL_dcov3D"]), _ptr(g["dL_dsh"]), _ptr(g["dL_dscales"]), _ptr(g["dL_drotations"]),
       _ptr(g["dL_dtau"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix, dtype=np.float32):
    dt = np.dtype(dtype)
    sfx = "f32" if dt == np.float32 else "f64"
    m = _arr(means3D, dt, empty_is_none=False)
    P = m.shape[0]
    present = np.zeros(P, np.uint8)
    if P:
        fn = getattr(lib(), f"gso_mark_visible_{sfx}")
        fn.restype = None
        fn(C.c_int(P), _ptr(m), _ptr(_arr(viewmatrix, dt)), _ptr(_arr(projmatrix, dt)), _ptr(present))
    return present.astype(bool)


def knn_dist2(points):
    """simple_knn.distCUDA2 restated (KNN/simple_knn.cu:147-183): mean squared distance to the 3 nearest others."""
    p = _arr(points, np.float32, empty_is_none=False).reshape(-1, 3)
    out = np.zeros(p.shape[0], np.float32)
    if p.shape[0]:
        fn = lib().gso_knn_dist2
        fn.restype = None
        fn(C.c_int(p.shape[0]), _ptr(p), _ptr(out))
    return out
