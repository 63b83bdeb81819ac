"""BackEnd.map_static() iteration time at SLAM scale (640x480, a map of a few 10k Gaussians, a full window of 8 keyframes + 2 random
ones per iteration), view by view (GSR_MULTI_VIEW=0) vs through the multi-view entry point; run once per setting:
    GSR_MULTI_VIEW=0 python tools/bench_backend_map.py ; python tools/bench_backend_map.py"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
from slam.camera import Camera
from slam.dataset import SyntheticRGBDDataset
from slam.system import SLAM, default_config, merge_config

torch.manual_seed(0)
n_kf = 11
ds = SyntheticRGBDDataset(num_frames=2 * n_kf, width=640, height=480, seed=0, spacing=0.025)
cfg = merge_config(default_config(), {"Training": {"init_itr_num": 300, "init_gaussian_reset": 150, "gaussian_update_every": 100000},
                                      "Dataset": {"pcd_downsample": 32, "pcd_downsample_init": 8}, "opt_params": {"densify_from_iter": 100}})
slam = SLAM(cfg, ds)
slam.frontend.run(max_frames=1)
fe, be = slam.frontend, slam.backend
for idx in range(2, 2 * n_kf, 2):
    cam = Camera.init_from_dataset(ds, idx, ds.projection_matrix)
    cam.compute_grad_mask(cfg)
    cam.update_RT(cam.R_gt, cam.T_gt)
    fe.cameras[idx] = cam
    be.viewpoints[idx] = cam
    be.add_next_kf(idx, cam, depth_map=fe.add_new_keyframe(idx))
    cam.reset_pose_optimizer()
window = [idx for idx in range(2 * n_kf - 2, 0, -2)][:8]
if "--eager" in sys.argv:
    cfg["Training"]["mapping_graph"] = False          # every iteration launched from Python (round 3's loop)
be.map_static(window, iters=10)
torch.cuda.synchronize()
ms0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
iters = 60
be.map_static(window, iters=iters)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters * 1e3
graph_stats = dict(getattr(be, "graph_stats", {}) or {})
# the replays alone: a second 60-iteration call minus its capture
c0 = graph_stats.get("capture_ms", 0.0)
t0 = time.perf_counter()
be.map_static(window, iters=iters)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) * 1e3
graph_stats2 = dict(getattr(be, "graph_stats", {}) or {})
replay_ms = (dt2 - (graph_stats2.get("capture_ms", 0.0) - c0)) / iters
ms1 = torch.cuda.memory_stats()
print({k: ms1[k] - ms0[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}, "reserved MB", ms1["reserved_bytes.all.current"] >> 20, file=sys.stderr)
from diff_gaussian_rasterization import _C
_C.profile_reset(); _C.profile_enable(True)
be.map_static(window, iters=5)
torch.cuda.synchronize(); _C.profile_enable(False)
kern = {k: round(v[0] / 5 * 1e3, 1) for k, v in _C.profile_read().items() if v[1]}
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); be.map_static(window, iters=20); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
print(json.dumps({"kernel_us_per_iteration": kern, "multi_view": os.environ.get("GSR_MULTI_VIEW", "1") != "0", "gaussians": int(be.gaussians.get_xyz.shape[0]), "views_per_iteration": len(window) + 2,
                  "ms_per_mapping_iteration": dt, "ms_per_mapping_iteration_without_capture": replay_ms, "graph_stats": graph_stats2,
                  "mode": "eager" if "--eager" in sys.argv else "graph"}))
