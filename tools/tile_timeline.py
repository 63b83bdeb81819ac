"""Residency timeline of the two tile kernels at BASELINE config #2 from a -DGSR_TIMELINE=1 build (per-block start / end on the chip-wide
100 MHz clock + HW_ID / XCC_ID):
    GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_variants/timeline.so python tools/tile_timeline.py [--json]
Answers, per kernel: how long the launch is from its first block's start to its last block's end, how many blocks are resident over time
(mean, and in deciles of the launch), how long a CU slot stays empty between two blocks (dispatch gap), how long the tail is (from the
moment the last block STARTS to the end), and whether the blocks that end last are the long ones."""
import ctypes, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench
from diff_gaussian_rasterization import _C


def _opt(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


P, W, H = _opt("--gaussians", 200_000, int), 640, 480
scene = bench.Scene(P, torch.device("cuda", 0), 0, _opt("--scale-mean", 0.005, float), keyframes=(0,))
for _ in range(5):
    scene.fwd_bwd(0)
torch.cuda.synchronize()
lib = _C.load_library()
if not hasattr(lib, "gsr_debug_spans"):
    raise SystemExit("this library was not built with -DGSR_TIMELINE=1")
lib.gsr_debug_spans.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]


def spans(which):
    buf = (ctypes.c_uint32 * (8192 * 4))()
    assert lib.gsr_debug_spans(buf, 8192 * 4, which) == 0
    a = np.frombuffer(buf, np.uint32).reshape(8192, 4).astype(np.int64)
    a = a[a[:, 1] != 0]
    return a


def analyse(a, name):
    t0 = a[:, 0].min()
    s, e = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01            # us
    dur = e - s
    span = e.max()
    grid = np.linspace(0, span, 201)
    resident = np.array([((s <= g) & (e > g)).sum() for g in grid])
    hw, xcc = a[:, 2], a[:, 3] & 0xF
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))   # (xcc, se, sh, cu)
    out = {"kernel": name, "blocks": int(len(a)), "first_start_to_last_end_us": round(float(span), 2),
           "block_us_mean_p50_p90_max": [round(float(x), 2) for x in (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max())],
           "last_block_starts_at_us": round(float(s.max()), 2), "tail_us": round(float(span - s.max()), 2),
           "mean_resident_blocks": round(float(dur.sum() / span), 1), "distinct_cus": int(len(np.unique(cu))),
           "resident_blocks_by_decile": [int(resident[int(i * 20 + 10)]) for i in range(10)],
           "starts_within_first_us": int((s < 1.0).sum())}
    # dispatch gap: per CU, sort blocks by start; greedy slots: a block starting after another block of the CU ended reuses its slot
    gaps = []
    per_cu = []
    for c in np.unique(cu):
        m = cu == c
        per_cu.append(int(m.sum()))
        ss, ee = s[m], e[m]
        order = np.argsort(ss)
        ends = []
        for i in order:
            free = [x for x in ends if x <= ss[i]]
            if free:
                x = max(free)
                gaps.append(ss[i] - x)
                ends.remove(x)
            ends.append(ee[i])
    out["blocks_per_cu_min_mean_max"] = [int(min(per_cu)), round(float(np.mean(per_cu)), 1), int(max(per_cu))]
    if gaps:
        g = np.array(gaps)
        out["slot_refill_gap_us_p50_p90_mean"] = [round(float(x), 2) for x in (np.percentile(g, 50), np.percentile(g, 90), g.mean())]
    last = np.argsort(e)[-max(1, len(e) // 20):]
    out["the_5pc_ending_last"] = {"mean_block_us": round(float(dur[last].mean()), 2), "mean_start_us": round(float(s[last].mean()), 2)}
    return out


if "--raw" in sys.argv:      # the raw records, for offline analysis (which CU took which block, in what order): index, start, end, HW_ID, XCC_ID
    dst = sys.argv[sys.argv.index("--raw") + 1]
    buf = (ctypes.c_uint32 * (8192 * 4))()
    raw = {}
    for which, name in ((0, "fwd"), (1, "bwd")):
        assert lib.gsr_debug_spans(buf, 8192 * 4, which) == 0
        raw[name] = np.frombuffer(buf, np.uint32).reshape(8192, 4).copy()
    np.savez_compressed(dst, **raw)
res = [analyse(spans(0), "render_fwd"), analyse(spans(1), "render_bwd")]
print(json.dumps(res, indent=1) if "--json" in sys.argv else "\n".join(str(r) for r in res))
