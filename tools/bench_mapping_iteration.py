#!/usr/bin/env python
"""One mapping iteration of the SLAM back-end (utils/slam_backend.py:336-771, single GPU) on synthetic data: K keyframes of the same
200k-Gaussian model are rendered (25 % dynamic Gaussians with control-node deltas), the mapping loss of every view is summed, one
backward, densification statistics per view, one Adam step.
  "reference_chain": what the reference executes around the rasterizer -- torch prologue, torch loss, boolean-mask statistics,
                     torch.optim.Adam -- over this repo's rasterizer;
  "fused":           prologue inside the kernels, fused loss, fused statistics, FusedAdam.
With --nodes the deltas of the dynamic Gaussians come from the SC-GS control-node warp (utils/time_utils.py:1192-1258: K = 3 nearest
of 512 nodes, RBF weights, local-frame blend) applied per view to leaf tensors standing for the node MLP's outputs -- the
reference's tensor program (brute-force cdist + topk for pytorch3d's knn_points) in "reference_chain", control_nodes.node_blend in
"fused" -- and their gradients flow back to the node attributes, radii and weights.
With --flow every view additionally renders the NDC flow towards the previous keyframe and back (render_flow twice, utils/slam_backend.py:
479-509) with an L1 loss against a stand-in flow field: the torch chain of the reference in "reference_chain", the fused route
(raw.rasterize_flow_raw) in "fused".
Prints one JSON line (profiles/r0N_mapping_iteration*.json)."""
import json, os, sys, time, types
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "4dgs-slam_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import gaussian_renderer as gr
from synthetic_scene import make_camera, make_gaussians, keyframe_pose
from synthetic_scene import GaussianModelStub as _GaussianModel, camera_namespace as _camera
from slam_losses import get_loss_mapping, mapping_loss_weights, add_densification_stats
from fused_adam import FusedAdam
import control_nodes as cn
from tools.bench_control_nodes import torch_program

NODES = "--nodes" in sys.argv
FLOW = "--flow" in sys.argv      # + the two render_flow calls per view of the dynamic mapping loop (utils/slam_backend.py:486,496)

P, W, H, K = 200_000, 640, 480, 8
config = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
g = make_gaussians(P, make_camera(W, H), seed=0, sh_degree=0)
rng = np.random.default_rng(11)
out = {"workload": f"{K} keyframes x {P} Gaussians @{W}x{H}, 25% dynamic with dx/ds/dr" + (" from the control-node warp (512 nodes, K=3)" if NODES else "")
                   + (", two render_flow per view" if FLOW else "") + ", mapping loss, densification stats, Adam"}
for fused in (False, True):
    m = _GaussianModel(g, False, 0.25, seed=2)
    m.max_radii2D = torch.zeros(P, device="cuda"); m.xyz_gradient_accum = torch.zeros(P, 1, device="cuda"); m.denom = torch.zeros(P, 1, device="cuda")
    views = []
    for k in range(K):
        R_w, t_w = keyframe_pose(k)
        v = _camera(make_camera(W, H, R=R_w, t=t_w))
        v.original_image = torch.tensor(rng.uniform(0, 1, size=(3, H, W)).astype(np.float32), device="cuda")
        v.depth, v.motion_mask, v.uid = rng.uniform(0.3, 5.0, size=(H, W)).astype(np.float32), None, k
        v.exposure_a = torch.nn.Parameter(torch.tensor([0.0], device="cuda")); v.exposure_b = torch.nn.Parameter(torch.tensor([0.0], device="cuda"))
        views.append(v)
    Kd = int(m.dygs.sum())
    deltas = [{k: torch.tensor(rng.normal(scale=s, size=(Kd, n)).astype(np.float32), device="cuda", requires_grad=True)
               for k, n, s in (("dx", 3, 0.002), ("ds", 3, 0.0001), ("dr", 4, 0.01))} for _ in range(K)]
    if NODES:
        M = 512
        T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
        nodes = T(g["means3D"][rng.choice(P, M, replace=False)])
        node_leaves = dict(rr=T(np.log(rng.uniform(0.2, 0.6, size=M)), True), wr=T(rng.normal(size=(M, 1)), True))
        per_view = [dict(tr=T(rng.normal(scale=0.002, size=(M, 3)), True), ro=T(rng.normal(scale=0.01, size=(M, 4)), True),
                         sc=T(rng.normal(scale=0.0001, size=(M, 3)), True), lr=T(rng.normal(scale=0.05, size=(M, 4)), True)) for _ in range(K)]
        motion = torch.ones(Kd, 1, device="cuda")

        def view_deltas(k):
            x = m._xyz.detach()[m.dygs]                            # gaussians.get_dygs_xyz.detach() (slam_backend.py:364)
            a = per_view[k]
            if fused:
                r = cn.node_blend(x, motion, nodes, node_leaves["rr"], node_leaves["wr"], a["tr"], a["ro"], a["sc"], a["lr"], K=3)
                return {"dx": r["d_xyz"], "ds": r["d_scaling"], "dr": r["d_rotation"]}
            dx, dr, ds = torch_program(x, motion, nodes, node_leaves["rr"], node_leaves["wr"], a["tr"], a["ro"], a["sc"], a["lr"], 3)
            return {"dx": dx, "ds": ds, "dr": dr}
    groups = [{"params": [p_], "lr": lr, "name": n} for n, p_, lr in (("xyz", m._xyz, 1.6e-4), ("f_dc", m._features_dc, 2.5e-3),
              ("opacity", m._opacity, 0.05), ("scaling", m._scaling, 1e-3), ("rotation", m._rotation, 1e-3))]
    opt = (FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    if fused and os.environ.get("GSR_FUSED_ACC", "1") == "1":      # the backward kernels add each view's gradients to one persistent buffer
        opt.enable_fused_gradient_accumulation()
    gr.FUSED_PROLOGUE = fused

    def torch_loss(image, depth, vp):
        gt_depth = torch.from_numpy(vp.depth).to(dtype=torch.float32, device=image.device)[None]
        w_rgb, w_dep = mapping_loss_weights(config, vp, vp.original_image, gt_depth)
        image_ab = torch.exp(vp.exposure_a) * image + vp.exposure_b
        return 0.9 * torch.abs(image_ab * w_rgb - vp.original_image * w_rgb).mean() + 0.1 * torch.abs(depth * w_dep - gt_depth * w_dep).mean()

    flow_target = torch.tensor(rng.normal(scale=0.01, size=(2, H, W)).astype(np.float32), device="cuda")
    os.environ["GSR_FUSED_FLOW"] = "1" if fused else "0"

    def iteration():
        opt.zero_grad(set_to_none=True)
        loss, pkgs = 0.0, []
        for k, (v, d) in enumerate(zip(views, deltas)):
            if NODES:
                d = view_deltas(k)
            res = gr.render(v, m, pipe, bg, **d)
            loss = loss + (get_loss_mapping(config, res["render"], res["depth"], v, res["opacity"]) if fused else torch_loss(res["render"], res["depth"], v))
            pkgs.append(res)
            if FLOW and k > 0:
                d2 = view_deltas(k - 1) if NODES else deltas[k - 1]
                f1 = gr.render_flow(m, v, views[k - 1], d["dx"], d2["dx"], d["dr"], d["ds"])
                f2 = gr.render_flow(m, views[k - 1], v, d2["dx"], d["dx"], d2["dr"], d2["ds"])
                loss = loss + 3 * torch.abs(flow_target - f1["render"][:2]).mean() + 3 * torch.abs(flow_target + f2["render"][:2]).mean()
        loss.backward()
        with torch.no_grad():
            for res in pkgs:
                if fused:
                    add_densification_stats(m, res["viewspace_points"], res["radii"])
                else:                                                              # utils/slam_backend.py:712-720, gaussian_model.py:973-977
                    vis = res["visibility_filter"]
                    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], res["radii"][vis])
                    m.xyz_gradient_accum[vis] += torch.norm(res["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)
                    m.denom[vis] += 1
        opt.step()

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out["fused" if fused else "reference_chain"] = {"ms_per_iteration": dt * 1e3, "gaussian_views_per_s": P * K / dt}
gr.FUSED_PROLOGUE = True
os.environ.pop("GSR_FUSED_FLOW", None)
out["speedup"] = out["reference_chain"]["ms_per_iteration"] / out["fused"]["ms_per_iteration"]
print(json.dumps(out))
