"""dev: per-wave cycle accounting of render_fwd (library built with -DGSR_FWD_TIMING=1, passed through GSR_LIB / GSR_GLUE=ctypes)."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench
from diff_gaussian_rasterization import _C

dev = torch.device("cuda", 0)
scene = bench.Scene(200_000, dev, 0, 0.005, keyframes=(0,))
for _ in range(5):
    scene.fwd_bwd(0)
torch.cuda.synchronize()
lib = _C.load_library()
T = 40 * 30
buf = (ctypes.c_uint32 * (T * 4 * 8))()
lib.gsr_debug_fwd_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.gsr_debug_fwd_timing(buf, T * 4 * 8) == 0
a = np.frombuffer(buf, np.uint32).reshape(T, 4, 8).astype(np.float64)
names = ["total", "sort", "stage", "list", "pair", "npairs", "batches", "wait"]
print("per-wave means:", {n: round(a[..., i].mean(), 1) for i, n in enumerate(names)})
tot = a[..., 0]
print("total cycles per wave: min %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(tot, [0, 50, 90, 99, 100])))
print("pairs per wave: p50 %.0f p90 %.0f max %.0f; cycles per pair (sum pair / sum npairs) %.1f" % (*np.percentile(a[..., 5], [50, 90, 100]), a[..., 4].sum() / a[..., 5].sum()))
tmax = tot.max(axis=1)
i = int(tmax.argmax())
print("slowest tile", i, {n: a[i, :, k].tolist() for k, n in enumerate(names)})
print("per-tile max total: p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(tmax, [50, 90, 100])))
# imbalance inside a tile: sum over batches is not available, but max/mean of pair cycles over the four quadrants is
pm = a[..., 4]
print("pair cycles: mean over waves %.0f, mean over tiles of max over quadrants %.0f" % (pm.mean(), pm.max(axis=1).mean()))

# ---- render_bwd ----
NC = 8192
buf = (ctypes.c_uint32 * (NC * 4 * 8))()
lib.gsr_debug_bwd_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.gsr_debug_bwd_timing(buf, NC * 4 * 8) == 0
b = np.frombuffer(buf, np.uint32).reshape(NC, 4, 8).astype(np.float64)
used = b[:, 0, 0] > 0
b = b[used]
names = ["total", "search", "state", "stage", "pair", "npairs", "epilogue", "t0"]
print("bwd chunks", len(b), "per-wave means:", {n: round(b[..., i].mean(), 1) for i, n in enumerate(names[:7])})
print("bwd cycles per pair %.1f; total per wave p50 %.0f p90 %.0f max %.0f" % (b[..., 4].sum() / b[..., 5].sum(), *np.percentile(b[..., 0], [50, 90, 100])))
t0 = b[..., 7]; end = t0 + b[..., 0]
print("bwd kernel span (cycles, wrapped 32-bit ok if < 2^32): %.0f" % (end.max() - t0.min()))

# ---- geometry_bwd ----
lib.gsr_debug_geo_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.gsr_debug_geo_timing(buf, NC * 4 * 8) == 0
g = np.frombuffer(buf, np.uint32).reshape(NC, 4, 8).astype(np.int64)
nb = (200_000 + 255) // 256
g = g[:nb]
d = np.diff(g[..., :7], axis=-1)
print("geometry_bwd blocks", nb, "phase means (meta+barrier, first slots, slot sums, param loads+math, stores, tau):", d.reshape(-1, 6).mean(0).round(0).tolist())
print("geometry_bwd: wave lifetime mean %.0f; kernel span %.0f cycles; start spread %.0f" % ((g[..., 6] - g[..., 0]).mean(), g[..., 6].max() - g[..., 0].min(), g[..., 0].max() - g[..., 0].min()))

# ---- preprocess_fwd / scatter_instances (1024-thread blocks: 16 waves) ----
lib.gsr_debug_pre_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
nb = (200_000 + 1023) // 1024
buf2 = (ctypes.c_uint32 * (2048 * 16 * 8))()
for which, name, nt in ((0, "preprocess_fwd (zero hist, per-Gaussian, histogram, block sum, row publish)", 6), (1, "scatter_instances (s_pos fill, prep, expand+write)", 4)):
    assert lib.gsr_debug_pre_timing(buf2, 2048 * 16 * 8, which) == 0
    q = np.frombuffer(buf2, np.uint32).reshape(2048, 16, 8).astype(np.int64)[:nb - 1]
    d = np.diff(q[..., :nt], axis=-1)
    print(name, "phase means:", d.reshape(-1, nt - 1).mean(0).round(0).tolist(), "wave lifetime %.0f" % (q[..., nt - 1] - q[..., 0]).mean(),
          "block span mean %.0f" % (q[..., nt - 1].max(axis=1) - q[..., 0].min(axis=1)).mean())
