import sys, ctypes, io, contextlib, runpy
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/4dgs-slam_amd")
from diff_gaussian_rasterization import _C
lib = _C.load_library()
out = np.zeros(8, np.uint64)
lib.gsr_debug_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
lib.gsr_debug_stats(out.ctypes.data, 0)
n = 1 + 5 + 0  # steps + breakdown passes (+1 no-grad forward does not run backward)
print("backward launches counted ~", n)
v, a, l, sb, rows = [float(x) for x in out[:5]]
print("pairs visited %.0f per launch; with any valid lane %.1f%%; valid lanes per processed pair %.1f of 64 (%.0f%%)" % (v / n, 100 * a / v, l / a, 100 * l / a / 64))
print("4x4 sub-blocks with a valid lane per processed pair: %.2f of 4; 8x2 rows: %.2f of 4" % (sb / a, rows / a))
