"""BASELINE.json configs[0]: 10k static Gaussians, 1 cam @320x240, CPU PyTorch fallback render + loss.backward() (plumbing, no
GPU). The "CPU PyTorch fallback rasterizer" is oracle/torch_raster.py (the reference has no CPU rasterizer);
it is driven through this repo's render() wrapper and checked against the C oracle. Test infrastructure only."""
import sys
import types

import numpy as np
import torch

from util import oracle_run, make_camera, make_gaussians, rel_l1
from oracle import torch_raster as tr


class _TorchRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.rs = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                theta=None, rho=None):
        rs = self.rs
        return tr.rasterize(means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                            cov3D_precomp=cov3D_precomp, bg=rs.bg, viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix, campos=rs.campos,
                            tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, image_height=rs.image_height, image_width=rs.image_width,
                            sh_degree=rs.sh_degree, scale_modifier=rs.scale_modifier)


def test_config1_cpu_render_and_backward(monkeypatch):
    import oracle.torch_binding as ob
    fake = types.ModuleType("diff_gaussian_rasterization")
    fake.GaussianRasterizationSettings = ob.GaussianRasterizationSettings
    fake.GaussianRasterizer = _TorchRasterizer
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", fake)
    for k in [k for k in sys.modules if k == "gaussian_renderer" or k.startswith("gaussian_renderer.")]:
        monkeypatch.delitem(sys.modules, k)
    import gaussian_renderer as gr

    torch.set_num_threads(8)
    W, H, P = 320, 240, 10_000
    c = make_camera(W, H)
    g = make_gaussians(P, c, seed=0)

    class Cam:
        image_height, image_width = H, W
        FoVx, FoVy = 2 * np.arctan(c.tanfovx), 2 * np.arctan(c.tanfovy)
        world_view_transform, full_proj_transform = torch.tensor(c.viewmatrix), torch.tensor(c.projmatrix)
        projection_matrix, camera_center = torch.tensor(c.projmatrix_raw), torch.tensor(c.campos)
        cam_rot_delta, cam_trans_delta = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
        time = 0.0

    class Model:
        get_xyz = torch.tensor(g["means3D"], requires_grad=True)
        get_features = torch.tensor(g["shs"], requires_grad=True)
        get_opacity = torch.tensor(g["opacities"], requires_grad=True)
        get_scaling = torch.tensor(g["scales"], requires_grad=True)
        get_rotation = torch.tensor(g["rotations"], requires_grad=True)
        dygs = torch.zeros(P, dtype=torch.bool)
        active_sh_degree = max_sh_degree = 0

    class Pipe:
        convert_SHs_python = compute_cov3D_python = False

    res = gr.render(Cam(), Model, Pipe(), torch.ones(3), dx=0, ds=0, dr=None)
    # mapping-style loss: L1 on colour + L1 on depth against constants (utils/slam_utils.py:274-364 shape)
    loss = 0.9 * (res["render"] - 0.5).abs().mean() + 0.1 * (res["depth"] - 2.0).abs().mean()
    loss.backward()
    for t in (Model.get_xyz, Model.get_features, Model.get_opacity, Model.get_scaling, Model.get_rotation, res["viewspace_points"]):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0
    oo, _, _ = oracle_run(g, c, np.ones(3, np.float32))
    assert rel_l1(res["render"].detach().numpy(), oo["color"]) < 1e-4
    assert rel_l1(res["depth"].detach().numpy(), oo["depth"]) < 1e-4
    assert (res["radii"].numpy() == oo["radii"]).all()
    assert (res["n_touched"].numpy() != oo["n_touched"]).sum() <= 2
