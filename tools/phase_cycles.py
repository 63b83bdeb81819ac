"""Per-wave cycle accounting of every kernel of the forward + backward step at BASELINE config #2 (200k Gaussians @640x480): where a wave's
lifetime goes, phase by phase (s_memtime ticks written by a library built with -DGSR_FWD_TIMING=1: tools/build_timing_lib.sh).
    GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py [--json]
Cycles are shader-clock cycles (~2.1 GHz on the boxes this ran on). This is what settled, in round 3, what bounds the kernels:
render_fwd's pair phase runs at ~2 cycles per wave-instruction per SIMD (issue-saturated: 4.7 waves per SIMD, all in the same phase),
its sort phase is bound by the per-wave issue rate of the sort networks; render_bwd's kernel time equals its pair count times the
mix-weighted issue cost of a pair; the per-Gaussian kernels spend their time in dependent global latencies."""
import ctypes, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench
from diff_gaussian_rasterization import _C

def _opt(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


# --gaussians / --scale-mean: another workload, e.g. a SLAM-sized map of few, large Gaussians (30000, 0.03: ~60 tiles per Gaussian)
P, W, H = _opt("--gaussians", 200_000, int), 640, 480
scene = bench.Scene(P, torch.device("cuda", 0), 0, _opt("--scale-mean", 0.005, float), keyframes=(0,))
for _ in range(5):
    scene.fwd_bwd(0)
torch.cuda.synchronize()
lib = _C.load_library()
if not hasattr(lib, "gsr_debug_fwd_timing"):
    raise SystemExit("this library was not built with -DGSR_FWD_TIMING=1 (tools/build_timing_lib.sh)")
out = {"workload": f"{P} Gaussians @{W}x{H}, one forward + backward; shader-clock cycles per wave (means over all waves of the launch)"}


def read(fn, n, *extra):
    buf = (ctypes.c_uint32 * n)()
    getattr(lib, fn).argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * len(extra)
    assert getattr(lib, fn)(buf, n, *extra) == 0
    return np.frombuffer(buf, np.uint32)


T = (W // 16) * (H // 16)
a = read("gsr_debug_fwd_timing", T * 4 * 8).reshape(T, 4, 8).astype(np.float64)
names = ["total", "sort", "stage", "sort_key_load", "pair_loops", "pairs", "sort_network", "sort_rank_gather_store"]
out["render_fwd"] = {n: round(float(a[..., i].mean()), 1) for i, n in enumerate(names)}
out["render_fwd"].update({"waves": int(T * 4), "cycles_per_pair_per_wave": round(float(a[..., 4].sum() / a[..., 5].sum()), 1),
                          "total_p50_p90_max": [float(x) for x in np.percentile(a[..., 0], [50, 90, 100])],
                          "pairs_per_wave_p50_p90_max": [float(x) for x in np.percentile(a[..., 5], [50, 90, 100])]})
NC = 8192
b = read("gsr_debug_bwd_timing", NC * 4 * 8).reshape(NC, 4, 8).astype(np.float64)
b = b[b[:, 0, 0] > 0]
names = ["total", "work_item_and_header", "pixel_state_loads", "staging", "pair_loops", "candidate_pairs", "group_epilogues"]
out["render_bwd"] = {n: round(float(b[..., i].mean()), 1) for i, n in enumerate(names)}
out["render_bwd"].update({"blocks": int(len(b)), "cycles_per_candidate_pair_per_wave": round(float(b[..., 4].sum() / b[..., 5].sum()), 1),
                          "total_p50_p90_max": [float(x) for x in np.percentile(b[..., 0], [50, 90, 100])]})
nb = (P + 255) // 256
g = read("gsr_debug_geo_timing", NC * 4 * 8).reshape(NC, 4, 8).astype(np.int64)[:nb]
d = np.diff(g[..., :7], axis=-1).reshape(-1, 6).mean(0)
out["geometry_bwd"] = dict(zip(["level1_loads_then_request_everything", "wait_for_slots_and_parameters_and_sum_slots", "wave_cooperative_gaussians", "chain_rules",
                                "pose_partial_sums", "stores"], [round(float(x), 1) for x in d]))
out["geometry_bwd"]["wave_lifetime"] = round(float((g[..., 6] - g[..., 0]).mean()), 1)
nb = (P + 1023) // 1024
for which, key, labels in ((0, "preprocess_fwd", ["zero_histogram", "per_gaussian", "tile_histogram", "block_sum", "publish_row"]),
                           (1, "scatter_instances", ["segment_cursors", "record_and_rect", "expand_and_write_keys"])):
    q = read("gsr_debug_pre_timing", 2048 * 16 * 8, which).reshape(2048, 16, 8).astype(np.int64)[:max(1, nb - 1)]
    nt = len(labels) + 1
    d = np.diff(q[..., :nt], axis=-1).reshape(-1, nt - 1).mean(0)
    out[key] = dict(zip(labels, [round(float(x), 1) for x in d]))
    out[key]["wave_lifetime"] = round(float((q[..., nt - 1] - q[..., 0]).mean()), 1)
if "--json" in sys.argv:
    print(json.dumps(out, indent=1))
else:
    for k, v in out.items():
        print(k, v)
