"""dev: render_fwd's batch loop in finer pieces, from a -DGSR_FWD_TIMING=2 build (see gs_render.h):
    GSR_GLUE=ctypes GSR_LIB=.../timing2.so python tools/dev_fwd_phases.py"""
import ctypes, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench
from diff_gaussian_rasterization import _C
P, W, H = 200_000, 640, 480
scene = bench.Scene(P, torch.device("cuda", 0), 0, 0.005, keyframes=(0,))
for _ in range(5):
    scene.fwd_bwd(0)
torch.cuda.synchronize()
lib = _C.load_library()
T = (W // 16) * (H // 16)
buf = (ctypes.c_uint32 * (T * 4 * 8))()
lib.gsr_debug_fwd_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.gsr_debug_fwd_timing(buf, T * 4 * 8) == 0
a = np.frombuffer(buf, np.uint32).reshape(T, 4, 8).astype(np.float64)
names = ["total", "sort", "stage", "barriers", "pair_loops", "pairs", "index_lists", "epilogue"]
out = {"mean_per_wave": {n: round(float(a[..., i].mean()), 1) for i, n in enumerate(names)}}
heavy = np.argsort(a[..., 0].max(1))[-60:]                      # the 5 % longest tiles
out["mean_per_wave_longest_5pc_tiles"] = {n: round(float(a[heavy][..., i].mean()), 1) for i, n in enumerate(names)}
blk = a[..., 0].max(1)
out["block_total_p50_p90_max"] = [float(x) for x in np.percentile(blk, [50, 90, 100])]
# per block: sum over waves of pair time vs 4 x the slowest wave's: how uneven the quadrants of a tile are
out["pair_loops_slowest_wave_over_mean_wave"] = round(float((a[..., 4].max(1) / np.maximum(a[..., 4].mean(1), 1)).mean()), 3)
print(json.dumps(out, indent=1))
