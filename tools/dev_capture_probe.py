"""Dev: which piece of the dynamic mapping iteration survives a hipGraph capture? usage: dev_capture_probe.py <piece>"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
from slam.deform_model import ControlNodes, DeformModel
piece = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
dm = DeformModel(device=dev)
pts = torch.rand(400, 3, device=dev)
dm.extend_node_from_point(pts)
nodes = dm.deform
x = torch.rand(5000, 3, device=dev)
tt = torch.linspace(0, 1, 30, device=dev)
w = torch.full((3,), 1e-3, device=dev)

def d_xyz_all(it):
    parts = [t for t in (it["d_xyz_full"], it["d_xyz_rest"]) if t is not None]
    return parts[0] if len(parts) == 1 else torch.cat(parts, 0)


def body():
    if piece == "trunk":
        it = nodes.begin_iteration_indexed(tt, 0)
        d_xyz_all(it).sum().backward()
    elif piece == "heads":
        it = nodes.begin_iteration_indexed(tt, 6)
        (d_xyz_all(it).sum() + sum(v.sum() for v in it["heads"].values())).backward()
    elif piece == "blend":
        it = nodes.begin_iteration_indexed(tt, 6, blend=(x, None))
        sum(r.sum() for rows in it["blended"] for r in rows).backward()
    elif piece == "elastic":
        it = nodes.begin_iteration_indexed(tt, 6)
        from slam.deform_model import elastic_error
        base = nodes.nodes.detach()
        e = base + it["d_xyz_rest"].reshape(2, 12, -1, 3)
        nn_weight, nn_idx = nodes._elastic_neighbours()
        elastic_error(e[:, 4:].permute(0, 2, 1, 3), nn_weight, nn_idx).sum().backward()
    elif piece == "arap":
        it = nodes.begin_iteration_indexed(tt, 6)
        from slam.deform_model import arap_error, connectivity_from_points
        base = nodes.nodes.detach()
        e = base + it["d_xyz_rest"].reshape(2, 12, -1, 3)
        seq = e[:, :4]
        nn_i, keep = connectivity_from_points(seq[:, 0], K=10)
        arap_error(seq, nn_i, keep).sum().backward()
    elif piece == "reg":
        it = nodes.begin_iteration_indexed(tt, 6)
        nodes.regularisers_indexed(it, 2, 0, w[:2]).backward()
    elif piece == "adam":
        it = nodes.begin_iteration_indexed(tt, 6)
        (d_xyz_all(it).sum() + sum(v.sum() for v in it["heads"].values())).backward()
        dm.optimizer.step()
        dm.optimizer.zero_grad(set_to_none=True)
    nodes.end_iteration()
    dm.optimizer.zero_grad(set_to_none=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    g.capture_begin()
    body()
    g.capture_end()
torch.cuda.current_stream().wait_stream(s2)
g.replay(); g.replay()
torch.cuda.synchronize()
print("PROBE", piece, "ok")
