#!/usr/bin/env python
"""Dev probe: which source lines of a DYNAMIC mapping iteration (executed directly, not replayed) issue which aten ops? A TorchDispatchMode
records every aten op of the calling thread with the innermost repo frame (the forward half and the updates); the backward half runs on the
autograd thread and is attributed by torch.profiler to its autograd node (tools/mapping_iteration_launches.py). usage: [--wh 640 480]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd")):
    sys.path.insert(0, p)
from slam.dataset import SyntheticRGBDDataset  # noqa: E402
from slam.system import SLAM, default_config, merge_config  # noqa: E402

wh = [int(v) for v in sys.argv[sys.argv.index("--wh") + 1:sys.argv.index("--wh") + 3]] if "--wh" in sys.argv else [640, 480]
torch.manual_seed(0)
ds = SyntheticRGBDDataset(num_frames=17, width=wh[0], height=wh[1], seed=0, dynamic=True, dystart=6, spacing=0.03)
cfg = merge_config(default_config(), {"Training": {"init_itr_num": 400, "init_gaussian_update": 100, "init_gaussian_reset": 200, "tracking_itr_num": 60,
                                                   "static_map_iters": 30, "dynamic_map_iters": 80, "network_init_iters": 50, "gaussian_update_every": 60,
                                                   "gaussian_update_offset": 20, "tracking_graph": True, "mapping_graph": False},
                                      "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 150},
                                      "model_params": {"dynamic_model": True}})
for i in range(len(ds)):
    ds[i]
slam = SLAM(cfg, ds)
slam.run()
be = slam.backend
window = list(be.current_window)
be.map(window, iters=3, dynamic_network=True)
torch.cuda.synchronize()
counts = collections.Counter()


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ("view", "reshape", "detach", "alias", "expand", "permute", "transpose", "select", "slice", "unsqueeze", "squeeze",
                                       "split", "as_strided", "t.default", "unbind", "_unsafe_view", "is_", "sym_", "stride", "size")):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "4dgs-slam_amd" in fr.filename and "tools/" not in fr.filename:
                    where = f"{fr.filename.split('4dgs-slam_amd/')[-1]}:{fr.lineno}"
                    break
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            counts[(where, name.replace("aten.", ""), str(shapes))] += 1
        return func(*args, **(kwargs or {}))


iters = 4
with Tracer():
    be.map(window, iters=iters, dynamic_network=True)
torch.cuda.synchronize()
for (where, op, shapes), c in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if c >= iters:
        print("%5.1f  %-48s %-32s %s" % (c / iters, where, op, shapes))
