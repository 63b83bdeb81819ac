import sys, os, ctypes, json, subprocess
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/4dgs-slam_amd")
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
from diff_gaussian_rasterization import _C
lib = _C.load_library()
T = 1200
t0 = np.zeros(T, np.uint64); t1 = np.zeros(T, np.uint64); n = np.zeros(T, np.int32)
lib.gsr_debug_tile_times.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
lib.gsr_debug_tile_times(t0.ctypes.data, t1.ctypes.data, n.ctypes.data, T)
d = (t1 - t0).astype(np.float64) / 100.0   # wall_clock64: 100 MHz -> us
start = (t0 - t0.min()).astype(np.float64) / 100.0
end = (t1 - t0.min()).astype(np.float64) / 100.0
print("tiles", T, "kernel span us %.1f" % end.max(), "mean tile us %.1f" % d.mean(), "max tile us %.1f" % d.max(), "latest start us %.1f" % start.max())
o = np.argsort(-d)[:8]
for i in o: print("tile", i, "n", n[i], "dur %.1f" % d[i], "start %.1f" % start[i], "us per entry %.3f" % (d[i] / max(n[i], 1)))
print("corr(n,dur)", np.corrcoef(n, d)[0, 1], "sum n", n.sum(), "max n", n.max())
print("us/entry overall: %.4f" % (d.sum() / n.sum()))
