#!/usr/bin/env python
"""Dev probe: which source lines of the reference-schedule stand-in run (tools/run_config4_stand_in.py's configuration) call the library's GEMMs
(aten mm / addmm / bmm / linear) on the calling thread, how often and with which shapes."""
import collections
import os
import sys
import tempfile
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset  # noqa: E402
from slam.system import SLAM, default_config, merge_config  # noqa: E402

torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=40, width=640, height=480, seed=0, dynamic=True, dystart=6, spacing=0.025)
cfg = merge_config(default_config(), {"Training": {"tracking_graph": True}, "model_params": {"dynamic_model": True}})
for i in range(len(ds)):
    ds[i]
counts = collections.Counter()


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ("aten.mm", "aten.addmm", "aten.bmm", "aten.baddbmm", "aten.linear", "aten.matmul")):
            stack = [f"{fr.filename.split('4dgs-slam_amd/')[-1]}:{fr.lineno}" for fr in traceback.extract_stack() if "4dgs-slam_amd" in fr.filename and "tools/" not in fr.filename]
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
            counts[(" < ".join(reversed(stack[-int(os.environ.get("DEPTH", "4")):])), name.replace("aten.", ""), str(shapes))] += 1
        return func(*args, **(kwargs or {}))


with tempfile.TemporaryDirectory() as tmp:
    slam = SLAM(cfg, ds, save_dir=tmp)
    with Tracer():
        slam.run(color_refinement_iters=200)
torch.cuda.synchronize()
for (where, op, shapes), c in sorted(counts.items(), key=lambda kv: -kv[1])[:40]:
    print("%6d  %-16s %-60s %s" % (c, op, shapes, where))
