"""dev: how the dispatcher of THIS box places render_fwd's blocks (from a -DGSR_TIMELINE=1 build): is block b on XCD b % 8, does an XCD deal its
blocks round robin over 32 CUs (k = b / 8 -> CU k % 32), how many blocks share a CU. One line of JSON.
    GSR_GLUE=ctypes GSR_LIB=.../timeline.so python tools/dev_dispatch_census.py"""
import ctypes, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench
from diff_gaussian_rasterization import _C
os.environ.setdefault("GSR_ORDER_TILES", "0")          # band order: tile index <-> block index by xcd_tile_of_block
scene = bench.Scene(200_000, torch.device("cuda", 0), 0, 0.005, keyframes=(0,))
lib = _C.load_library()
lib.gsr_debug_spans.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
T = 1200
q, r = T >> 3, T & 7
def block_of_tile(i):
    head = r * (q + 1)
    if i < head:
        return (i % (q + 1)) * 8 + i // (q + 1)
    rest = i - head
    return (rest % q) * 8 + r + rest // q
blk = np.array([block_of_tile(t) for t in range(T)])
res = []
for frame in range(3):
    scene.fwd_bwd(0)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint32 * (8192 * 4))()
    assert lib.gsr_debug_spans(buf, 8192 * 4, 0) == 0
    f = np.frombuffer(buf, np.uint32).reshape(8192, 4)[:T].astype(np.int64)
    hw, xcc = f[:, 2], f[:, 3] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    k = blk // 8
    rr = np.mean([cu[np.where((blk % 8 == x) & (k == kk))[0][0]] == cu[np.where((blk % 8 == x) & (k == kk + 32))[0][0]] for x in range(8) for kk in range(0, 96)])
    fill = np.mean([cu[np.where((blk % 8 == x) & (k == kk))[0][0]] == cu[np.where((blk % 8 == x) & (k == kk + 1))[0][0]] for x in range(8) for kk in range(0, 140)])
    counts = np.bincount(np.unique(cu, return_counts=True)[1], minlength=8).tolist()
    dur = (f[:, 1] - f[:, 0]) * 0.01
    res.append({"xcd_is_block_mod_8": float((xcc == blk % 8).mean()), "same_cu_as_k_plus_32": float(rr), "same_cu_as_k_plus_1": float(fill), "distinct_cus": int(len(np.unique(cu))),
                "cus_by_blocks_hosted": counts, "first_xcd_first_cus": [int(c & 0xFF) for c in cu[np.argsort(blk)][0:64:8]], "fwd_span_us": float((f[:, 1].max() - f[:, 0].min()) * 0.01),
                "block_us_p50_max": [float(np.percentile(dur, 50)), float(dur.max())]})
print(json.dumps(res))
