"""GPU parity of the fused HexPlane field / deformation network (include/deformation_field.h) against
  * the golden vectors of the reference's own HexPlaneField and deform_network (tests/golden/make_golden_deformation.py),
  * the pinned CPU oracle (oracle/deformation_oracle.py) in fp64 on seeded inputs at the shipped geometry,
  * torch's own F.grid_sample composition on the GPU (an independent fp32 statement of the same op).
Tolerances: values rel-L1 <= 1e-5 (north_star asks 1e-4), gradients rel-L1 <= 1e-4 (north_star asks 1e-3)."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import deformation
import hexplane
from oracle import deformation_oracle as O
from test_deformation_host import hidden_params

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_deformation.npz"))
DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-30))


def torch_field(pts, tim, aabb, levels):
    """F.grid_sample statement of HexPlaneField.forward (what the reference executes), on whatever device."""
    p = torch.clamp((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, -1.0, 1.0)
    p4 = torch.cat((p, tim), dim=-1)
    feats = []
    for planes in levels:
        prod = 1.0
        for (c0, c1), plane in zip(itertools.combinations(range(4), 2), planes):
            grid = p4[:, [c0, c1]].view(1, 1, -1, 2)
            s = F.grid_sample(plane, grid, align_corners=True, mode="bilinear", padding_mode="border")   # [1, C, 1, n]
            prod = prod * s.view(plane.shape[1], -1).t()
        feats.append(prod)
    return torch.cat(feats, dim=-1)


def golden_field(layout):
    levels = []
    for l in range(2):
        lv = []
        for p in range(6):
            t = torch.tensor(G[f"field/plane_{l}_{p}"], device=DEV)
            t = t.contiguous(memory_format=torch.channels_last) if layout == "channels_last" else t.contiguous()
            lv.append(t.requires_grad_(True))
        levels.append(lv)
    return levels


@pytest.mark.parametrize("layout", ["channels_last", "contiguous", "channels_last_binned"])
def test_field_matches_reference_golden(layout, monkeypatch):
    monkeypatch.setenv("GSR_HEX_BINNED", "1" if layout.endswith("binned") else "0")
    layout = layout.replace("_binned", "")
    levels = golden_field(layout)
    pts = torch.tensor(G["field/pts"], device=DEV, requires_grad=True)
    feat = hexplane.hexplane_features(pts, torch.tensor(G["field/time"], device=DEV), torch.tensor(G["field/aabb"], device=DEV), levels)
    assert feat.shape == (256, 64)
    assert rel(feat, G["field/features"]) < 1e-6
    (feat * torch.tensor(G["field/cotangent"], device=DEV)).sum().backward()
    assert rel(pts.grad, G["field/g_pts"]) < 1e-5
    for l in range(2):
        for p in range(6):
            g = levels[l][p].grad
            assert g.shape == levels[l][p].shape and g.stride() == levels[l][p].stride()
            assert rel(g, G[f"field/g_plane_{l}_{p}"]) < 1e-5, (l, p)
    # exact zeros where the reference has exact zeros (clamped / border coordinates)
    assert torch.all((pts.grad.cpu() == 0) == torch.tensor(G["field/g_pts"] == 0))


@pytest.mark.parametrize("fused_mlp", [True, False])
@pytest.mark.parametrize("tag", ["net1", "net2"])
def test_network_matches_reference_golden(tag, fused_mlp, monkeypatch):
    """net1 (defor_depth 1) takes the fused MLP kernels (csrc/gs_mlp.h) when fused_mlp, net2 always the layer-by-layer path."""
    monkeypatch.setattr(deformation, "FUSED_MLP", fused_mlp)
    res = [int(r) for r in G[f"{tag}/resolution"]]
    net = deformation.deform_network(hidden_params(defor_depth=int(G[f"{tag}/defor_depth"]), multires=[int(m) for m in G[f"{tag}/multires"]],
                                                   kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4,
                                                                   "output_coordinate_dim": 32, "resolution": res}), DEV).to(DEV)
    net.load_state_dict({k: torch.tensor(G[f"{tag}/state/{k}"]) for k in G[f"{tag}/state_keys"]}, strict=True)
    ins = {k: torch.tensor(G[f"{tag}/in_{k}"], device=DEV) for k in ("point", "scales", "rotations", "opacity", "shs", "time")}
    for k in ("point", "scales", "rotations"):
        ins[k].requires_grad_(True)
    outs = net(ins["point"], ins["scales"], ins["rotations"], ins["opacity"], ins["shs"], ins["time"])
    loss = 0
    for name, o in zip(["means3D", "scales", "rotations", "dx", "ds", "dr"], outs):
        assert rel(o, G[f"{tag}/out_{name}"]) < 1e-5, name
        loss = loss + (o * torch.tensor(G[f"{tag}/cot_{name}"], device=DEV)).sum()
    loss.backward()
    for k in ("point", "scales", "rotations"):
        assert rel(ins[k].grad, G[f"{tag}/g_in_{k}"]) < 1e-4, k
    for name, p in net.named_parameters():
        g = G[f"{tag}/grad/{name}"]
        if g.size:
            assert rel(p.grad, g) < 1e-4, name


def default_field(seed=0, scale=1.0):
    torch.manual_seed(seed)
    field = hexplane.HexPlaneField(1.6, {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                         "resolution": [64, 64, 64, 25]}, [1, 2, 4, 8]).to(DEV)
    with torch.no_grad():
        for lv in field.grids:
            for p in lv:
                p.uniform_(0.2, 1.3)                                 # informative time planes (they start as ones)
    return field


@pytest.mark.parametrize("binned", ["0", "1"])
def test_shipped_geometry_against_oracle_fp64_and_torch_grid_sample(binned, monkeypatch):
    """binned = "1": the counting-sort + LDS-accumulation backward (gs_hexplane_binned.h); "0": one atomic per (point, corner)."""
    monkeypatch.setenv("GSR_HEX_BINNED", binned)
    field = default_field()
    rng = np.random.default_rng(3)
    n = 20000
    pts = torch.tensor(rng.uniform(-1.9, 1.9, size=(n, 3)).astype(np.float32), device=DEV, requires_grad=True)
    tim = torch.tensor(rng.uniform(-1.1, 1.1, size=(n, 1)).astype(np.float32), device=DEV)
    cot = torch.tensor(rng.normal(size=(n, 128)).astype(np.float32), device=DEV)
    feat = field(pts, tim)
    (feat * cot).sum().backward()
    g_pts = pts.grad.clone()
    g_planes = [[p.grad.clone() for p in lv] for lv in field.grids]
    # (a) torch's grid_sample on the GPU, fp32, all points
    pts2 = pts.detach().clone().requires_grad_(True)
    lv2 = [[p.detach().clone().contiguous().requires_grad_(True) for p in lv] for lv in field.grids]
    feat2 = torch_field(pts2, tim, field.aabb, lv2)
    (feat2 * cot).sum().backward()
    assert rel(feat, feat2) < 1e-6
    assert rel(g_pts, pts2.grad) < 1e-5
    for l in range(4):
        for p in range(6):
            assert rel(g_planes[l][p], lv2[l][p].grad) < 1e-5, (l, p)
    # (b) the pinned oracle in fp64 on a subset (values; the plane gradients of a subset need their own backward)
    m = 1500
    sub = pts.detach()[:m].double().cpu().requires_grad_(True)
    lv64 = [[p.detach().double().cpu().contiguous().requires_grad_(True) for p in lv] for lv in field.grids]
    f64 = O.hexplane_field(sub, tim[:m].double().cpu(), field.aabb.detach().double().cpu(), lv64)
    assert rel(feat[:m], f64) < 2e-5        # fp32 product of six fp32 samples against the exact answer
    (f64 * cot[:m].double().cpu()).sum().backward()
    pts3 = pts.detach()[:m].clone().requires_grad_(True)
    for lv in field.grids:
        for p in lv:
            p.grad = None
    (field(pts3, tim[:m]) * cot[:m]).sum().backward()
    assert rel(pts3.grad, sub.grad) < 1e-4
    for l in range(4):
        for p in range(6):
            assert rel(field.grids[l][p].grad, lv64[l][p].grad) < 1e-4, (l, p)   # fp32 kernel vs the exact answer


@pytest.mark.parametrize("binned", ["0", "1"])
@pytest.mark.parametrize("C", [8, 16, 64])
def test_other_feature_widths(C, binned, monkeypatch):
    monkeypatch.setenv("GSR_HEX_BINNED", binned)
    rng = np.random.default_rng(C)
    res = [5, 4, 6, 3]
    levels = []
    for m in (1, 3):
        r = [res[0] * m, res[1] * m, res[2] * m, res[3]]
        lv = [torch.tensor(rng.uniform(0.2, 1.2, size=(1, C, r[c1], r[c0])).astype(np.float32), device=DEV)
              .contiguous(memory_format=torch.channels_last).requires_grad_(True) for c0, c1 in itertools.combinations(range(4), 2)]
        levels.append(lv)
    n = 777
    pts = torch.tensor(rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32), device=DEV, requires_grad=True)
    tim = torch.tensor(rng.uniform(-1.0, 1.0, size=(n, 1)).astype(np.float32), device=DEV)
    aabb = torch.tensor([[1.0, 1.1, 0.9], [-1.0, -0.8, -1.2]], device=DEV)
    feat = hexplane.hexplane_features(pts, tim, aabb, levels)
    cot = torch.tensor(rng.normal(size=(n, 2 * C)).astype(np.float32), device=DEV)
    (feat * cot).sum().backward()
    p64 = pts.detach().double().cpu().requires_grad_(True)
    l64 = [[p.detach().double().cpu().contiguous().requires_grad_(True) for p in lv] for lv in levels]
    f64 = O.hexplane_field(p64, tim.double().cpu(), aabb.double().cpu(), l64)
    (f64 * cot.double().cpu()).sum().backward()
    assert rel(feat, f64) < 2e-5 and rel(pts.grad, p64.grad) < 1e-4
    for l in range(2):
        for p in range(6):
            assert rel(levels[l][p].grad, l64[l][p].grad) < 1e-4


def test_edge_cases_empty_ragged_strided_and_expanded():
    field = default_field(1)
    # empty: the reference returns zeros [0, 1] (hexplane.py:174-175)
    out = field(torch.zeros((0, 3), device=DEV), torch.zeros((0, 1), device=DEV))
    assert out.shape == (0, 1)
    rng = np.random.default_rng(5)
    for n in (1, 7, 33, 257):
        emb = torch.tensor(rng.uniform(-1.5, 1.5, size=(n, 63)).astype(np.float32), device=DEV, requires_grad=True)
        t0 = torch.tensor([[0.25]], device=DEV)
        a = field(emb[:, :3], t0.expand(n, 1))                       # strided rows (the reference's rays_pts_emb[:, :3]) + stride-0 time
        b = field(emb[:, :3].detach().contiguous(), t0.repeat(n, 1))
        assert torch.equal(a, b)
        a.sum().backward()
        assert emb.grad.shape == (n, 63) and float(emb.grad[:, 3:].abs().max()) == 0.0
    # NaN / inf coordinates must not fault (ATen clips them onto the border)
    bad = torch.tensor([[float("nan"), 0.0, 0.0], [float("inf"), -float("inf"), 0.0]], device=DEV)
    out = field(bad, torch.zeros((2, 1), device=DEV))
    torch.cuda.synchronize()
    assert out.shape == (2, 128)


def test_interpolate_ms_features_entry_point():
    levels = golden_field("channels_last")
    aabb = torch.tensor(G["field/aabb"], device=DEV)
    pts = torch.tensor(G["field/pts"], device=DEV)
    tim = torch.tensor(G["field/time"], device=DEV)
    p4 = torch.cat((hexplane.normalize_aabb(pts, aabb), tim), dim=-1)
    f = hexplane.interpolate_ms_features(p4, levels, grid_dimensions=2, concat_features=True, num_levels=None)
    assert rel(f, G["field/features"]) < 1e-6
    s = hexplane.interpolate_ms_features(p4, levels, grid_dimensions=2, concat_features=False, num_levels=None)
    assert rel(s, G["field/features"][:, :32] + G["field/features"][:, 32:]) < 1e-6
    one = hexplane.interpolate_ms_features(p4, levels, 2, True, num_levels=1)
    assert rel(one, G["field/features"][:, :32]) < 1e-6


def test_bad_arguments_raise():
    levels = golden_field("channels_last")
    pts = torch.zeros((4, 3), device=DEV)
    tim = torch.zeros((4, 1), device=DEV)
    with pytest.raises(ValueError, match="inconsistent resolution"):
        bad = [list(levels[0]), list(levels[1])]
        bad[0][1] = levels[1][1]
        hexplane.hexplane_features(pts, tim, None, bad)
    with pytest.raises(ValueError, match="six planes"):
        hexplane.hexplane_features(pts, tim, None, [levels[0][:5]])
    with pytest.raises(ValueError, match="share one memory layout"):
        mixed = [list(levels[0])]
        mixed[0][3] = levels[0][3].detach().contiguous()
        hexplane.hexplane_features(pts, tim, None, mixed)
    with pytest.raises(ValueError, match="timestamps"):
        hexplane.hexplane_features(pts, torch.zeros((3, 1), device=DEV), None, levels)
    with pytest.raises(Exception, match="HIP device|no CPU"):
        hexplane.hexplane_features(pts.cpu(), tim.cpu(), None, levels)


def test_render_dynamic_through_the_deformation_network():
    """render(dynamic=True) (gaussian_renderer/__init__.py:149-157) with the build's deform_network as pc._deformation: gradients
    reach the planes, the MLP and the Gaussian parameters -- and the route that hands the network's outputs to the kernels as raw
    parameters (exp / normalize in-kernel) gives the same image and gradients as the reference's torch chain."""
    import types
    import gaussian_renderer
    from util import make_camera, make_gaussians
    from test_hip_fused_prologue import _GaussianModel, _camera
    cam = make_camera(160, 120)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    torch.manual_seed(0)
    net = deformation.deform_network(hidden_params(multires=[1, 2], bounds=8.0), DEV).to(DEV)
    results = {}
    for fused in (False, True):
        pc = _GaussianModel(make_gaussians(3000, cam, seed=2), isotropic=False, dyn_frac=0.0, seed=3)
        view = _camera(cam)
        view.time = 0.4
        pc._deformation = net
        for p in net.parameters():
            p.grad = None
        gaussian_renderer.FUSED_PROLOGUE = fused
        try:
            assert gaussian_renderer._fused_prologue_ok(pc, pipe, None, True) == fused
            out = gaussian_renderer.render(view, pc, pipe, bg, dynamic=True)
        finally:
            gaussian_renderer.FUSED_PROLOGUE = True
        (out["render"].mean() + 0.1 * out["depth"].mean()).backward()
        grads = {"xyz": pc._xyz.grad, "scaling": pc._scaling.grad, "rotation": pc._rotation.grad, "opacity": pc._opacity.grad,
                 "theta": view.cam_rot_delta.grad, "rho": view.cam_trans_delta.grad}
        grads.update({"net." + k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
        results[fused] = (out, grads)
    (o0, g0), (o1, g1) = results[False], results[True]
    assert rel(o1["render"], o0["render"]) < 1e-5 and rel(o1["depth"], o0["depth"]) < 1e-5 and torch.equal(o1["radii"], o0["radii"])
    assert set(g0) == set(g1)
    for k in g0:
        if float(g0[k].abs().max()) < 1e-12:
            assert float(g1[k].abs().max()) < 1e-9, k
        else:
            assert rel(g1[k], g0[k]) < 5e-4, (k, rel(g1[k], g0[k]))
    grid_g = [v for k, v in g1.items() if "grids" in k]
    assert len(grid_g) == 12 and sum(float(g.abs().sum()) for g in grid_g) > 0
    assert float(g1["net.deformation_net.pos_deform.3.weight"].abs().sum()) > 0 and torch.isfinite(g1["xyz"]).all()


@pytest.mark.parametrize("n,in_dim,out_dim", [(1, 64, 3), (5, 4, 64), (63, 128, 64), (64, 64, 4), (1000, 100, 48), (4097, 128, 70),
                                              (200003, 128, 64), (200003, 64, 3)])
def test_linear_weight_gradient_kernel(n, in_dim, out_dim):
    """gsr_linear_wgrad (split-K fp32 MFMA) against the same products in fp64; strided rows; deterministic."""
    torch.manual_seed(n + in_dim)
    wide = torch.randn(n, in_dim + 7, device=DEV)
    x = wide[:, 3:3 + in_dim]                                           # row stride in_dim + 7
    lin = deformation.PointwiseLinear(in_dim, out_dim).to(DEV)
    y = lin(x.requires_grad_(False))
    gy = torch.randn(n, out_dim, device=DEV)
    y.backward(gy)
    gw64 = gy.double().t() @ x.double()
    gb64 = gy.double().sum(0)
    scale = (gy.double().abs().t() @ x.double().abs()).clamp_min(1e-30)   # sum |a b|: the natural unit of the rounding error
    assert float(((lin.weight.grad.double() - gw64).abs() / scale).max()) < 2e-6
    assert float(((lin.bias.grad.double() - gb64).abs() / gy.double().abs().sum(0).clamp_min(1e-30)).max()) < 2e-6
    g1 = lin.weight.grad.clone()
    lin.weight.grad = None
    lin.bias.grad = None
    lin(x).backward(gy)
    assert torch.equal(g1, lin.weight.grad)                            # fixed summation order
    # value and input gradient are the library GEMM
    x2 = x.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.linear(x2, lin.weight, lin.bias)
    assert torch.allclose(lin(x2), ref)


@pytest.mark.parametrize("case", ["clustered", "one_texel", "uniform_t_large", "border"])
def test_binned_backward_on_hostile_distributions(case, monkeypatch):
    """The binned backward against the direct one: every point in one bin (split over several blocks), every point in ONE texel,
    a large uniform-time batch (the render(dynamic=True) call shape), points on / outside the aabb faces."""
    field = default_field(4)
    rng = np.random.default_rng(9)
    n = 60000
    if case == "clustered":
        pts = rng.normal(scale=0.01, size=(n, 3)) + np.array([0.3, -0.4, 0.2])
        tim = np.full((n, 1), 0.1)
    elif case == "one_texel":
        pts = np.tile(np.array([[0.123, 0.456, -0.789]]), (n, 1)) + rng.uniform(-1e-5, 1e-5, size=(n, 3))
        tim = np.full((n, 1), -0.5)
    elif case == "uniform_t_large":
        n = 150000
        pts = rng.uniform(-1.6, 1.6, size=(n, 3))
        tim = np.full((n, 1), 0.73)
    else:
        pts = rng.choice(np.array([-1.7, -1.6, 1.6, 1.7, 0.0]), size=(n, 3))
        tim = rng.choice(np.array([-1.2, -1.0, 1.0, 1.3]), size=(n, 1))
    pts_t = torch.tensor(pts.astype(np.float32), device=DEV)
    tim_t = torch.tensor(tim.astype(np.float32), device=DEV)
    cot = torch.tensor(rng.normal(size=(n, 128)).astype(np.float32), device=DEV)
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSR_HEX_BINNED", mode)
        for lv in field.grids:
            for p in lv:
                p.grad = None
        x = pts_t.clone().requires_grad_(True)
        (field(x, tim_t) * cot).sum().backward()
        results[mode] = (x.grad.clone(), [[p.grad.clone() for p in lv] for lv in field.grids])
    assert torch.equal(results["0"][0], results["1"][0])                      # dL/dxyz does not go through the atomics
    for l in range(4):
        for p in range(6):
            a, b = results["0"][1][l][p], results["1"][1][l][p]
            # both sum the same terms in different orders: compare against the size of the terms
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) / scale < 2e-4, (case, l, p, float((a - b).abs().max()) / scale)
            assert rel(a, b) < 1e-4, (case, l, p)


@pytest.mark.parametrize("n,feat_levels", [(1, 4), (17, 2), (1000, 4), (4099, 1)])
def test_fused_mlp_against_layerwise_path(n, feat_levels, monkeypatch):
    """the fused MLP kernels (fp32 MFMA) against the same network evaluated layer by layer with library GEMMs: values, input
    gradient and every weight / bias gradient; n not a multiple of the 16-point tile, in_dim 32 / 64 / 128."""
    torch.manual_seed(n)
    net = deformation.deform_network(hidden_params(multires=[1, 2, 4, 8][:feat_levels],
                                                   kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                                                   "resolution": [8, 8, 8, 5]}), DEV).to(DEV)
    with torch.no_grad():
        for p in net.parameters():
            if p.requires_grad and p.dim() == 1:
                p.normal_(0, 0.2)                                    # non-trivial biases
    rng = np.random.default_rng(n)
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=DEV, requires_grad=rg)
    ins = [T(rng.uniform(-1.5, 1.5, size=(n, 3)), True), T(rng.normal(size=(n, 3)), True), T(rng.normal(size=(n, 4)), True),
           T(rng.normal(size=(n, 1))), T(rng.normal(size=(n, 16, 3))), T(np.full((n, 1), 0.2))]
    cots = [T(rng.normal(size=(n, c))) for c in (3, 3, 4, 3, 3, 4)]
    results = {}
    for fused in (False, True):
        monkeypatch.setattr(deformation, "FUSED_MLP", fused)
        for p in net.parameters():
            p.grad = None
        for t in ins[:3]:
            t.grad = None
        outs = net(*ins)
        sum((o * c).sum() for o, c in zip(outs, cots)).backward()
        results[fused] = ([o.detach().clone() for o in outs], [t.grad.clone() for t in ins[:3]],
                          {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    (o0, gi0, gp0), (o1, gi1, gp1) = results[False], results[True]
    for a, b in zip(o1, o0):
        assert rel(a, b) < 1e-5
    for a, b in zip(gi1, gi0):
        assert rel(a, b) < 1e-4
    assert set(gp0) == set(gp1)
    for k in gp0:
        assert rel(gp1[k], gp0[k]) < 1e-4, (k, rel(gp1[k], gp0[k]))


def test_points_without_gradient_are_skipped_correctly(monkeypatch):
    """Gaussians a view does not see arrive with an exactly-zero cotangent: the sorted field backward leaves them out of the sort and
    the fused MLP backward skips whole 64-point tiles of them.  Same gradients as the direct / layer-by-layer paths, zero dL/dxyz."""
    torch.manual_seed(1)
    net = deformation.deform_network(hidden_params(), DEV).to(DEV)
    with torch.no_grad():
        for p in net.get_grid_parameters():
            if p.requires_grad:
                p.uniform_(0.2, 1.2)
    rng = np.random.default_rng(4)
    n = 60000
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=DEV, requires_grad=rg)
    ins = [T(rng.uniform(-1.5, 1.5, size=(n, 3)), True), T(rng.normal(size=(n, 3)), True), T(rng.normal(size=(n, 4)), True),
           T(rng.normal(size=(n, 1))), None, T(np.full((n, 1), -0.3))]
    seen = rng.uniform(size=n) < 0.3                                  # scattered visible points ...
    seen[20000:45000] = False                                         # ... and a long run of invisible ones (whole MLP tiles)
    seen_t = torch.tensor(seen, device=DEV)[:, None]
    cots = [T(rng.normal(size=(n, c))) * seen_t for c in (3, 3, 4)]
    results = {}
    for mode in ("reference_paths", "fast_paths"):
        monkeypatch.setenv("GSR_HEX_BINNED", "0" if mode == "reference_paths" else "1")
        monkeypatch.setattr(deformation, "FUSED_MLP", mode == "fast_paths")
        for p in net.parameters():
            p.grad = None
        for t in ins[:3]:
            t.grad = None
        outs = net(*ins)
        sum((o * c).sum() for o, c in zip(outs[3:], cots)).backward()      # only dx, ds, dr carry a cotangent: invisible rows stay exactly zero
        results[mode] = (ins[0].grad.clone(), {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    (gx0, gp0), (gx1, gp1) = results["reference_paths"], results["fast_paths"]
    assert torch.all(gx1[~seen_t[:, 0]] == 0) and rel(gx1, gx0) < 1e-4
    assert set(gp0) == set(gp1)
    for k in gp0:
        assert rel(gp1[k], gp0[k]) < 2e-4, (k, rel(gp1[k], gp0[k]))


# ---- the views of one mapping iteration at once (gsr_hexplane_*_views, deform_network.forward_views, render_views(dynamic=True)) --------
def _shipped_field(bounds=1.6, multires=(1, 2, 4, 8), seed=0):
    torch.manual_seed(seed)
    field = hexplane.HexPlaneField(bounds, {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]},
                                   list(multires)).to(DEV)
    with torch.no_grad():
        for lv in field.grids:
            for p in lv:
                p.copy_(torch.empty_like(p).uniform_(0.2, 1.2))        # the time planes start at exactly 1: give them structure
    return field


@pytest.mark.parametrize("n,V", [(1, 1), (3000, 8), (70000, 12), (60000, 2)])
def test_batched_time_field_equals_per_view_field(n, V, monkeypatch):
    """HexPlaneField.forward_views (spatial planes gathered once, one sort + one spatial scatter for all views) against V calls of the
    single-view field at the same points: values BIT-identical (the product keeps the reference's plane order), plane and position
    gradients <= 1e-5 rel-L1 (sums regroup). Points outside the aabb, times outside [-1, 1], whole zero cotangent rows (Gaussians a view
    does not see) included."""
    field = _shipped_field()
    g = torch.Generator(device="cpu").manual_seed(n + V)
    pts = (torch.rand((n, 3), generator=g) * 4.0 - 2.0).to(DEV).requires_grad_(True)
    times = [float(t) for t in np.linspace(-1.2, 1.2, V)] if V > 1 else [0.3]
    cot = torch.randn((V, n, field.feat_dim), generator=g).to(DEV)
    cot[:, torch.rand(n, generator=g).to(DEV) < 0.2] = 0.0                 # points no view sees
    if V > 1:
        cot[1, torch.rand(n, generator=g).to(DEV) < 0.5] = 0.0             # ... and points one view does not see
    batched = field.forward_views(pts, times)
    assert batched is not None and batched.shape == (V, n, field.feat_dim)
    (batched * cot).sum().backward()
    gb = {"pts": pts.grad.clone(), **{k: p.grad.clone() for k, p in field.named_parameters() if p.grad is not None}}
    pts.grad = None
    for p in field.parameters():
        p.grad = None
    loss = 0.0
    for v, t in enumerate(times):
        f_v = field(pts, torch.full((n, 1), t, device=DEV))
        assert torch.equal(f_v, batched[v]), v
        loss = loss + (f_v * cot[v]).sum()
    loss.backward()
    gs = {"pts": pts.grad, **{k: p.grad for k, p in field.named_parameters() if p.grad is not None}}
    assert set(gb) == set(gs) and len(gs) == 1 + 24
    for k in gs:
        assert gb[k].shape == gs[k].shape and gb[k].stride() == gs[k].stride(), k
        assert rel(gb[k], gs[k]) < 1e-5, (k, rel(gb[k], gs[k]))
    assert torch.equal(gb["pts"] == 0, gs["pts"] == 0)                    # exact zeros where the clamp / the border stop the gradient


def test_batched_time_field_reproduces_reference_golden():
    """golden_deformation.npz (the reference's own HexPlaneField) through the batched entry point, as a one-view and a two-view call."""
    levels = golden_field("channels_last")
    pts = torch.tensor(G["field/pts"], device=DEV, requires_grad=True)
    time = torch.tensor(G["field/time"], device=DEV)
    if float(time.min()) != float(time.max()):
        pytest.skip("the golden call uses per-point times")
    t = float(time[0])
    feat = hexplane.hexplane_features_views(pts, [t, t], torch.tensor(G["field/aabb"], device=DEV), levels)
    assert rel(feat[0], G["field/features"]) < 1e-6 and torch.equal(feat[0], feat[1])
    (feat[0] * torch.tensor(G["field/cotangent"], device=DEV)).sum().backward()
    assert rel(pts.grad, G["field/g_pts"]) < 1e-5
    for l in range(2):
        for p in range(6):
            assert rel(levels[l][p].grad, G[f"field/g_plane_{l}_{p}"]) < 1e-5, (l, p)


def test_batched_field_rejects_what_it_does_not_cover():
    field = _shipped_field(multires=(1,))
    pts = torch.rand((10, 3), device=DEV)
    with pytest.raises(ValueError):
        hexplane.hexplane_features_views(pts, [0.0] * 13, field.aabb, hexplane._PlaneList(field.grids))
    planar = [[p.detach().contiguous() for p in lv] for lv in field.grids]
    assert not hexplane.views_supported(planar, 2) and hexplane.views_supported(hexplane._PlaneList(field.grids), 2)
    out = hexplane.hexplane_features_views(pts, [0.1, 0.2], field.aabb, planar)      # the forward handles the reference's layout too
    for v, t in enumerate((0.1, 0.2)):
        assert torch.equal(out[v], hexplane.hexplane_features(pts, torch.full((10, 1), t, device=DEV), field.aabb, planar))


@pytest.mark.parametrize("n,V,zero_frac", [(1, 1, 0.0), (5000, 3, 0.6), (40000, 12, 0.9), (3000, 8, 1.0)])
def test_network_views_with_and_without_the_row_mask(n, V, zero_frac, monkeypatch):
    """deform_network.forward_views (ONE autograd node: field of all views + MLP over V * n rows). Its backward lists the rows of the
    cotangent that are not zero (gsr_row_mask), runs the MLP's backward over the list (gsr_deform_mlp_backward_rows) and lets the field skip the
    others by their view bit: same gradients as with every row processed (GSR_ROW_MASK=0), which in turn are those of V forward_dynamic
    calls. zero_frac 1.0: no row at all is listed."""
    torch.manual_seed(1)
    net = deformation.deform_network(hidden_params(multires=[1, 2], bounds=1.6), DEV).to(DEV)
    with torch.no_grad():
        for p_ in net.get_grid_parameters():
            p_.copy_(torch.empty_like(p_).uniform_(0.2, 1.2))
    g = torch.Generator(device="cpu").manual_seed(7 * n + V)
    base = (torch.rand((n, 3), generator=g) * 3.6 - 1.8).to(DEV)
    times = [float(t) for t in np.linspace(-0.9, 0.9, V)] if V > 1 else [0.25]
    cot = torch.randn((V, n, 10), generator=g).to(DEV)
    cot[(torch.rand((V, n), generator=g) < zero_frac).to(DEV)] = 0.0
    if n > 100:
        cot[:, :70] = 0.0                                                     # whole 64-row tiles of the list's neighbourhood, points no view lists
    results = {}
    for mode in ("1", "0", "per_view"):
        monkeypatch.setenv("GSR_ROW_MASK", "0" if mode == "0" else "1")
        pts = base.clone().requires_grad_(True)
        for p_ in net.parameters():
            p_.grad = None
        if mode == "per_view":
            zeros3, zeros4 = torch.zeros((n, 3), device=DEV), torch.zeros((n, 4), device=DEV)
            outs = []
            for t in times:
                _, _, _, dx, ds, dr = net(pts, zeros3, zeros4, None, None, torch.full((n, 1), t, device=DEV))
                outs.append(torch.cat((dx, ds, dr), dim=1))
            out = torch.stack(outs)
        else:
            out = net.forward_views(pts, times)
        assert out is not None and out.shape == (V, n, 10)
        (out * cot).sum().backward()
        results[mode] = (out.detach(), {"pts": pts.grad.clone(), **{k: p_.grad.clone() for k, p_ in net.named_parameters() if p_.grad is not None}})
    assert torch.equal(results["1"][0], results["0"][0]) and rel(results["1"][0], results["per_view"][0]) < 1e-6
    for other, tol in (("0", 2e-6), ("per_view", 1e-4)):
        ga, gb = results["1"][1], results[other][1]
        assert set(ga) == set(gb)
        for k in gb:
            if float(gb[k].abs().max()) == 0.0:
                assert float(ga[k].abs().max()) == 0.0, (other, k)
            else:
                assert rel(ga[k], gb[k]) < tol, (other, k, rel(ga[k], gb[k]))


def test_row_mask_lists_the_nonzero_rows_in_order():
    """gsr_row_mask against torch: view bits per point, the flat indices of the non-zero rows in ascending order, their count on the device."""
    import ctypes
    lib = deformation._lib()
    for V, n, width in ((1, 1, 10), (3, 1000, 10), (12, 70001, 10), (32, 257, 4)):
        g = torch.randn((V, n, width), device=DEV)
        g[torch.rand((V, n), device=DEV) < 0.7] = 0.0
        g[0, 0] = 0.0
        g[V - 1, n - 1, width - 1] = 1.0                                       # a row whose only non-zero element is its last
        mask = torch.empty((n,), dtype=torch.int32, device=DEV)
        rows = torch.full((V * n + 1,), -7, dtype=torch.int32, device=DEV)
        ws = torch.empty((lib.gsr_row_mask_workspace_size(V, n),), dtype=torch.uint8, device=DEV)
        rc = lib.gsr_row_mask(V, n, width, g.data_ptr(), mask.data_ptr(), rows.data_ptr(), rows[V * n:].data_ptr(), ws.data_ptr(), None)
        assert rc == 0
        nz = (g != 0).any(dim=-1)                                               # [V, n]
        want = nz.flatten().nonzero().flatten().to(torch.int32)
        assert int(rows[V * n]) == want.numel() and torch.equal(rows[:want.numel()], want)
        bits = sum((nz[v].to(torch.int64) << v) for v in range(V))
        assert torch.equal(mask.to(torch.int64) & 0xFFFFFFFF, bits)
    assert lib.gsr_row_mask(33, 10, 10, g.data_ptr(), mask.data_ptr(), rows.data_ptr(), rows.data_ptr(), ws.data_ptr(), None) < 0


@pytest.mark.parametrize("row_mask", ["1", "0"])
@pytest.mark.parametrize("iso", [False, True])
def test_render_views_dynamic_equals_per_camera_render_dynamic(iso, row_mask, monkeypatch):
    """render_views(dynamic=True): ONE evaluation of the deformation network for all cameras' times + the multi-view rasterizer with the
    network's output as deltas in front of the activations (gsr_raw_inputs.delta_mode = 1, delta_stride = 10) against one
    render(dynamic=True) per camera, with the backward's row list on and off. Run twice: the first call of a view slot goes through the
    single-view kernels inside gsr_forward_views, the second through the batched ones."""
    monkeypatch.setenv("GSR_ROW_MASK", row_mask)
    import types
    import gaussian_renderer
    from synthetic_scene import keyframe_pose
    from util import make_camera, make_gaussians
    from test_hip_fused_prologue import _GaussianModel, _camera
    W, H, V, P = 160, 120, 4, 3000
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    torch.manual_seed(0)
    net = deformation.deform_network(hidden_params(multires=[1, 2], bounds=8.0), DEV).to(DEV)
    with torch.no_grad():
        for p_ in net.get_grid_parameters():
            p_.copy_(torch.empty_like(p_).uniform_(0.05, 0.3))
    cam0 = make_camera(W, H)

    def scene():
        pc = _GaussianModel(make_gaussians(P, cam0, seed=2), isotropic=iso, dyn_frac=0.0, seed=3)
        pc._deformation = net
        views = []
        for k in range(V):
            R_w, t_w = keyframe_pose(k)
            v = _camera(make_camera(W, H, R=R_w, t=t_w))
            v.time = k / (V - 1) * 1.6 - 0.8
            views.append(v)
        return pc, views

    def grads_of(pc, views, outs):
        for p in net.parameters():
            p.grad = None
        loss = sum((o["render"] * (1.0 + 0.1 * k)).mean() + 0.1 * o["depth"].mean() for k, o in enumerate(outs))
        loss.backward()
        g = {"xyz": pc._xyz.grad, "scaling": pc._scaling.grad, "rotation": pc._rotation.grad, "opacity": pc._opacity.grad, "f_dc": pc._features_dc.grad}
        for k, v in enumerate(views):
            g[f"theta{k}"], g[f"rho{k}"], g[f"m2d{k}"] = v.cam_rot_delta.grad, v.cam_trans_delta.grad, outs[k]["viewspace_points"].grad
        g.update({"net." + k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
        return g

    pc, views = scene()
    ref_outs = [gaussian_renderer.render(v, pc, pipe, bg, dynamic=True) for v in views]
    ref = grads_of(pc, views, ref_outs)
    for attempt in range(2):
        pc, views = scene()
        outs = gaussian_renderer.render_views(views, pc, pipe, bg, dynamic=True)
        assert isinstance(outs[0], gaussian_renderer._RenderPackage)             # the batched route was taken
        for o, r in zip(outs, ref_outs):
            assert torch.equal(o["radii"], r["radii"]) and torch.equal(o["n_touched"], r["n_touched"])
            assert rel(o["render"], r["render"]) < 1e-6 and rel(o["depth"], r["depth"]) < 1e-6 and rel(o["opacity"], r["opacity"]) < 1e-6
        got = grads_of(pc, views, outs)
        assert set(got) == set(ref)
        for k in ref:
            if float(ref[k].abs().max()) < 1e-12:
                assert float(got[k].abs().max()) < 1e-9, k
            else:
                assert rel(got[k], ref[k]) < (1e-4 if k.startswith("net.") else 2e-5), (attempt, k, rel(got[k], ref[k]))


def _plane_grads(field):
    return [p.grad.clone() for lv in field.grids for p in lv]


@pytest.mark.parametrize("route", ["views", "single"])
def test_plane_gradients_are_bitwise_reproducible_and_do_not_depend_on_the_point_order(route, monkeypatch):
    """Ordered mode (gsr_set_option "hex_ordered", the default; gs_hexplane_binned.h HexOrd): every dL/dsample x corner-weight product is rounded
    once to a power-of-two quantum and summed as integers, so the plane gradients are the same BITS (a) run to run and (b) for any
    permutation of the points -- the order the counting sort's cursors or the atomics happen to take cannot matter. The float-atomic mode
    of rounds 1-5 agrees to rounding."""
    from diff_gaussian_rasterization import _C
    assert _C.set_option("hex_ordered") == 1
    monkeypatch.setenv("GSR_HEX_BINNED", "1")
    field = _shipped_field(seed=5)
    n, V = 40000, 5
    g = torch.Generator(device="cpu").manual_seed(11)
    pts = ((torch.rand((n, 3), generator=g) * 3.6 - 1.8) * torch.tensor([1.0, 0.3, 0.05])).to(DEV)       # crowded along y and z: long runs, busy cells
    times = [float(t) for t in np.linspace(-0.9, 1.1, V)]
    cot = torch.randn((V, n, field.feat_dim), generator=g).to(DEV) * torch.logspace(-6, 2, n).to(DEV)[None, :, None]
    cot[:, torch.rand(n, generator=g).to(DEV) < 0.3] = 0.0
    cot[2, torch.rand(n, generator=g).to(DEV) < 0.5] = 0.0

    def run(perm=None):
        for p in field.parameters():
            p.grad = None
        x, c = (pts, cot) if perm is None else (pts[perm], cot[:, perm])
        x = x.clone().requires_grad_(True)
        if route == "views":
            (field.forward_views(x, times) * c).sum().backward()
        else:
            (field(x, torch.full((n, 1), times[1], device=DEV)) * c[1]).sum().backward()
        return _plane_grads(field), x.grad

    base, gx = run()
    assert all(torch.isfinite(b).all() for b in base) and sum(float(b.abs().sum()) for b in base) > 0
    for _ in range(3):
        again, gx2 = run()
        assert all(torch.equal(a, b) for a, b in zip(again, base)) and torch.equal(gx, gx2)
    perm = torch.randperm(n, generator=g).to(DEV)
    shuffled, gxp = run(perm)
    assert all(torch.equal(a, b) for a, b in zip(shuffled, base))
    assert torch.equal(gxp, gx[perm])
    # the float-atomic mode: the same sums to rounding
    old = _C.set_option("hex_ordered", 0)
    try:
        loose, _ = run()
    finally:
        _C.set_option("hex_ordered", old)
    for a, b in zip(loose, base):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-30
        assert rel(a, b) < 1e-5


def test_ordered_plane_gradients_scale_with_the_cotangent():
    """The fixed-point quantum follows the call's largest |dL/dsample|: cotangents scaled by 2^-60 or 2^+40 give gradients scaled by exactly
    that power of two (no overflow, no flush to zero)."""
    field = _shipped_field(seed=6)
    n, V = 20000, 3
    g = torch.Generator(device="cpu").manual_seed(3)
    pts = (torch.rand((n, 3), generator=g) * 3.0 - 1.5).to(DEV)
    cot = torch.randn((V, n, field.feat_dim), generator=g).to(DEV)
    out = {}
    for k in (0, -60, 40):
        for p in field.parameters():
            p.grad = None
        (field.forward_views(pts, [-0.5, 0.0, 0.8]) * (cot * 2.0 ** k)).sum().backward()
        out[k] = _plane_grads(field)
    for k in (-60, 40):
        for a, b in zip(out[k], out[0]):
            assert torch.equal(a, b * 2.0 ** k)


def test_ordered_plane_gradients_beyond_a_million_points():
    """n > 2^20: the fixed-point budget shrinks below 40 bits (62 - ceil(log2 4n), gs_capi.hip hexord_plan) so that 4 n contributions of the
    largest magnitude cannot overflow 63 bits. Every point in ONE texel with the same sign -- the worst case for the sum --: finite,
    reproducible, and equal to the float-atomic mode's sums to its own rounding noise."""
    from diff_gaussian_rasterization import _C
    field = _shipped_field(seed=7)
    n = 1_300_000
    g = torch.Generator(device="cpu").manual_seed(2)
    pts = (torch.tensor([[0.123, 0.456, -0.789]]) + (torch.rand((n, 3), generator=g) - 0.5) * 1e-4).to(DEV)
    cot = (torch.rand((1, n, field.feat_dim), generator=g) + 0.5).to(DEV)             # all positive: nothing cancels

    def run():
        for p in field.parameters():
            p.grad = None
        (field.forward_views(pts, [0.25]) * cot).sum().backward()
        return _plane_grads(field)

    a = run()
    b = run()
    assert all(torch.isfinite(x).all() for x in a) and all(torch.equal(x, y) for x, y in zip(a, b))
    old = _C.set_option("hex_ordered", 0)
    try:
        loose = run()
    finally:
        _C.set_option("hex_ordered", old)
    for x, y in zip(a, loose):
        assert float(x.abs().max()) > 0
        assert float((x - y).abs().max()) <= 1e-4 * float(y.abs().max())           # (650 k float additions into one texel: the atomic mode's own drift)


@pytest.mark.parametrize("poison", [float("nan"), float("inf")])
def test_ordered_plane_gradients_stay_loud_about_non_finite_cotangents(poison, monkeypatch):
    """A NaN / infinite cotangent cannot be represented in the fixed-point sums; instead of dropping it the ordered mode makes EVERY plane
    gradient of the call NaN (the float path would have delivered the value to the texels it touches)."""
    monkeypatch.setenv("GSR_HEX_BINNED", "1")
    field = _shipped_field(seed=8)
    n = 5000
    g = torch.Generator(device="cpu").manual_seed(4)
    pts = (torch.rand((n, 3), generator=g) * 3.0 - 1.5).to(DEV)
    for route in ("views", "single"):
        cot = torch.randn((2, n, field.feat_dim), generator=g).to(DEV)
        cot[1, 1234, 17] = poison
        for p in field.parameters():
            p.grad = None
        if route == "views":
            (field.forward_views(pts, [-0.5, 0.4]) * cot).sum().backward()
        else:
            (field(pts, torch.full((n, 1), 0.4, device=DEV)) * cot[1]).sum().backward()
        grads = _plane_grads(field)
        assert len(grads) == 24 and all(torch.isnan(x).all() for x in grads), route
