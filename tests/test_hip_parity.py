"""GPU parity tests: the HIP product path (through the C ABI, via the drop-in Python package) against the oracle on the
same seeded inputs. Bars: RGB / depth / opacity rel-L1 <= 1e-4, every gradient tensor rel-L1 <= 1e-3,
radii / visibility / n_touched exact up to boundary flips (counted). All marked gpu."""
import os

import numpy as np
import pytest
import torch

from util import (oracle, oracle_run, hip_run, compare, make_camera, make_gaussians, make_cotangents, keyframe_pose, rel_l1)

pytestmark = pytest.mark.gpu

IMG_TOL, GRAD_TOL = 1e-4, 1e-3


def _check(m, flips=0, nt_tol=2):
    for k in ("color", "depth", "opacity"):
        assert m[k] <= IMG_TOL, (k, m)
    for k, v in m.items():
        if k.startswith("g_"):
            assert v <= GRAD_TOL, (k, m)
    assert m["radii_mismatch"] <= flips and m["visible_mismatch"] <= flips, m
    assert m["n_touched_gt0_mismatch"] <= flips, m
    # n_touched counts pixels with T(1-alpha) > 0.5: a handful of exact-threshold pixels may flip between exp() implementations
    assert m["n_touched_mismatch"] <= max(flips, nt_tol + m.get("P", 0) // 50000), m


CASES = [
    # P, W, H, deg, max_deg, scale_mean, kw
    (2000, 160, 120, 0, 0, 0.005, {}),
    (2000, 150, 100, 3, 3, 0.01, {}),                 # W, H not multiples of 16; full SH
    (3000, 96, 64, 1, 3, 0.02, {}),                   # active degree < allocated coefficients (M = 16, D = 1)
    (5000, 64, 48, 0, 0, 0.05, {}),                   # long per-tile lists, heavy saturation
    (20000, 640, 480, 2, 2, 0.005, {}),
    (4000, 200, 120, 0, 0, 0.01, {"precomp_color": True}),
    (4000, 200, 120, 0, 0, 0.01, {"precomp_cov": True}),
    (3000, 128, 96, 1, 1, 0.01, {"scale_modifier": 0.6}),
    (3000, 128, 96, 0, 0, 0.01, {"keyframe": 7}),     # rotated + translated camera (pose gradient through a non-identity W2C)
]


@pytest.mark.parametrize("P,W,H,deg,maxdeg,sm,kw", CASES)
def test_forward_backward_parity(P, W, H, deg, maxdeg, sm, kw):
    R, t = keyframe_pose(kw.get("keyframe", 0))
    cam = make_camera(W, H, R=R, t=t)
    g = make_gaussians(P, make_camera(W, H), seed=P % 17, sh_degree=deg, max_sh_degree=maxdeg, scale_mean=sm)
    if kw.get("keyframe"):
        g["means3D"] = ((g["means3D"].astype(np.float64) - t) @ R).astype(np.float32)
    gc, gd = make_cotangents(cam)
    bg = np.array([1.0, 0.5, 0.2], np.float32)
    cp = np.random.default_rng(5).uniform(-1, 1, (P, 3)).astype(np.float32) if kw.get("precomp_color") else None
    cov = None
    if kw.get("precomp_cov"):
        _, st0, _ = oracle_run(g, cam, bg)
        cov = st0.state()["cov3D"]
    smod = kw.get("scale_modifier", 1.0)
    oo, st, go = oracle_run(g, cam, bg, gc, gd, colors_precomp=cp, cov3D_precomp=cov, scale_modifier=smod)
    oh, gh = hip_run(g, cam, bg, gc, gd, colors_precomp=cp, cov3D_precomp=cov, scale_modifier=smod)
    m = compare(oh, gh, oo, go)
    m["P"] = P
    _check(m)


def test_intermediate_state_parity():
    """Stage-by-stage: geometry state, scan, per-tile ranges and the depth-sorted lists must match the reference pipeline
    (preprocess / InclusiveSum / duplicateWithKeys + SortPairs / identifyTileRanges)."""
    from diff_gaussian_rasterization import _C
    cam = make_camera(200, 136)
    P = 6000
    g = make_gaussians(P, cam, seed=2, sh_degree=2, scale_mean=0.01)
    bg = np.zeros(3, np.float32)
    oo, st, _ = oracle_run(g, cam, bg)
    so = st.state()
    T = lambda a: torch.tensor(a, device="cuda")
    nr, color, radii, gb, bb, ib, depth, opac, nt = _C.rasterize_gaussians(
        T(bg), T(g["means3D"]), torch.Tensor([]), T(g["opacities"]), T(g["scales"]), T(g["rotations"]), 1.0, torch.Tensor([]),
        T(cam.viewmatrix), T(cam.projmatrix), T(cam.projmatrix_raw), cam.tanfovx, cam.tanfovy, cam.H, cam.W, T(g["shs"]), 2, T(cam.campos),
        False, False)
    assert nr == oo["num_rendered"]
    sh = _C.debug_read_state(P, nr, cam.W, cam.H, gb, bb, ib)
    vis = oo["radii"] > 0
    assert (radii.cpu().numpy() == oo["radii"]).all()
    assert (sh["tiles_touched"] == so["tiles_touched"]).all() and (sh["point_offsets"] == so["point_offsets"]).all()
    for k, tol in (("depths", 1e-6), ("means2D", 1e-5), ("conic_opacity", 2e-5), ("rgb", 1e-5)):
        assert rel_l1(sh[k][vis], so[k][vis]) < tol, k
    inf = so["depths"] > 0.2   # cov3D is written for everything in front of the near plane
    assert rel_l1(sh["cov3D"][inf], so["cov3D"][inf]) < 1e-5
    assert (sh["clamped"][vis] == so["clamped"][vis]).all()
    assert (sh["ranges"] == so["ranges"]).all()
    assert (sh["point_list"] == so["point_list"]).all()          # identical order, ties included
    assert (sh["n_contrib"] == so["n_contrib"]).mean() > 0.999
    assert rel_l1(sh["final_T"], so["final_T"]) < 1e-4


def test_depth_ties_keep_index_order():
    from diff_gaussian_rasterization import _C
    cam = make_camera(64, 48)
    P = 500
    rng = np.random.default_rng(0)
    g = make_gaussians(P, cam, seed=9, scale_mean=0.03)
    g["means3D"][:, 2] = np.repeat(rng.uniform(1, 3, 10), 50).astype(np.float32)   # only 10 distinct depths
    g["means3D"][:, :2] *= 0.3
    bg = np.zeros(3, np.float32)
    gc, gd = make_cotangents(cam)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    oh, gh = hip_run(g, cam, bg, gc, gd)
    _check(compare(oh, gh, oo, go))


@pytest.mark.parametrize("P", [9000, 21000])
def test_huge_tile_list_uses_global_sort_path(P):
    """More than SORT_LDS_CAP (4096) instances in one tile: chunk-wise LDS sort + rank by counting (3 and 6 chunks; the last one partial)."""
    cam = make_camera(24, 16)           # 2 tiles, the left one holds most of the Gaussians
    g = make_gaussians(P, cam, seed=3, scale_mean=0.004)
    g["means3D"][::7, 2] = g["means3D"][3, 2]                  # many equal depths: ties must keep the instance order across chunks
    g["means3D"][:, 0] = -np.abs(g["means3D"][:, 0]) * 0.6
    g["opacities"][:] = 0.02                                   # keep transmittance alive through the whole list
    gc, gd = make_cotangents(cam)
    bg = np.ones(3, np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    assert (st.state()["ranges"][:, 1] - st.state()["ranges"][:, 0]).max() > 4096
    oh, gh = hip_run(g, cam, bg, gc, gd)
    _check(compare(oh, gh, oo, go))


def test_edge_cases():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = make_camera(70, 50)
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
    rs = lambda **o: GaussianRasterizationSettings(**{**dict(
        image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=T([0.3, 0.6, 0.9]), scale_modifier=1.0,
        viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw), sh_degree=0, campos=T(cam.campos),
        prefiltered=False, debug=False), **o})
    # empty model: outputs stay zero (rasterize_points.cu:85), gradients are empty tensors
    m3 = torch.zeros((0, 3), device="cuda", requires_grad=True)
    c, r, d, o, n = GaussianRasterizer(rs())(means3D=m3, means2D=torch.zeros((0, 3), device="cuda"), opacities=torch.zeros((0, 1), device="cuda"),
                                             shs=torch.zeros((0, 1, 3), device="cuda"), scales=torch.zeros((0, 3), device="cuda"),
                                             rotations=torch.zeros((0, 4), device="cuda"))
    assert c.shape == (3, cam.H, cam.W) and float(c.abs().max()) == 0.0 and r.shape == (0,)
    # everything behind the near plane: pure background, zero gradients, nothing rendered
    P = 100
    g = make_gaussians(P, cam, seed=1)
    g["means3D"][:, 2] = 0.1
    oh, gh = hip_run(g, cam, np.array([0.3, 0.6, 0.9], np.float32), *make_cotangents(cam))
    np.testing.assert_allclose(oh["color"], np.broadcast_to(np.array([0.3, 0.6, 0.9], np.float32)[:, None, None], oh["color"].shape), rtol=1e-6)
    assert (oh["radii"] == 0).all() and all(np.abs(v).max() == 0 for v in gh.values() if v is not None)
    # prefiltered + culled point is an error (the reference traps the device, auxiliary.h:156-160)
    with pytest.raises(RuntimeError, match="prefiltered"):
        GaussianRasterizer(rs(prefiltered=True))(means3D=T(g["means3D"]), means2D=T(np.zeros((P, 3))), opacities=T(g["opacities"]),
                                                  shs=T(g["shs"]), scales=T(g["scales"]), rotations=T(g["rotations"]))
    # markVisible == near-plane test
    g2 = make_gaussians(P, cam, seed=2)
    g2["means3D"][::2, 2] = 0.15
    vis = GaussianRasterizer(rs()).markVisible(T(g2["means3D"])).cpu().numpy()
    assert (vis == oracle.mark_visible(g2["means3D"], cam.viewmatrix, cam.projmatrix)).all() and vis.sum() == P // 2
    # debug=True path (sync after every stage) gives the same image
    oh2, _ = hip_run(make_gaussians(500, cam, seed=5), cam, np.zeros(3, np.float32))
    c2, *_ = GaussianRasterizer(rs(bg=T([0, 0, 0]), debug=True))(
        means3D=T(make_gaussians(500, cam, seed=5)["means3D"]), means2D=T(np.zeros((500, 3))), opacities=T(make_gaussians(500, cam, seed=5)["opacities"]),
        shs=T(make_gaussians(500, cam, seed=5)["shs"]), scales=T(make_gaussians(500, cam, seed=5)["scales"]),
        rotations=T(make_gaussians(500, cam, seed=5)["rotations"]))
    assert np.array_equal(c2.cpu().numpy(), oh2["color"])


def test_backward_is_bit_reproducible_and_linear():
    """No float atomics anywhere: two runs agree bit for bit. The backward is linear in the cotangents."""
    cam = make_camera(160, 120)
    g = make_gaussians(5000, cam, seed=4, sh_degree=1)
    bg = np.ones(3, np.float32)
    gc1, gd1 = make_cotangents(cam, seed=1)
    gc2, gd2 = make_cotangents(cam, seed=2)
    _, a = hip_run(g, cam, bg, gc1, gd1)
    _, a2 = hip_run(g, cam, bg, gc1, gd1)
    _, b = hip_run(g, cam, bg, gc2, gd2)
    _, c = hip_run(g, cam, bg, 2 * gc1 - 3 * gc2, 2 * gd1 - 3 * gd2)
    for k in a:
        if a[k] is not None:
            assert np.array_equal(a[k], a2[k]), k
            assert rel_l1(c[k], 2 * a[k] - 3 * b[k]) < 2e-5, k


def test_full_size_config2_properties_and_subsampled_parity():
    """BASELINE.json configs[1]: 200k Gaussians @640x480. Full oracle comparison (the C oracle needs ~5 s) plus the
    size-independent invariants: opacity = 1 - final_T in [0, 0.9999], colour within [0, max], sorted lists, permutation invariance."""
    from diff_gaussian_rasterization import _C
    cam = make_camera(640, 480)
    P = 200_000
    g = make_gaussians(P, cam, seed=0)
    gc, gd = make_cotangents(cam)
    bg = np.ones(3, np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    oh, gh = hip_run(g, cam, bg, gc, gd)
    m = compare(oh, gh, oo, go)
    m["P"] = P
    _check(m, flips=2)
    assert oh["opacity"].min() >= 0 and oh["opacity"].max() <= 1 - 1e-4 + 1e-6
    mse = float(((oh["color"] - oo["color"]) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 80        # PSNR vs oracle image
    # permuting the Gaussians changes nothing but the order of equal-depth ties
    perm = np.random.default_rng(0).permutation(P)
    gp = {k: (v[perm] if isinstance(v, np.ndarray) and v.shape[:1] == (P,) else v) for k, v in g.items()}
    op, _ = hip_run(gp, cam, bg)
    assert rel_l1(op["color"], oh["color"]) < 1e-5 and (op["radii"] == oh["radii"][perm]).all()


def test_transposed_wave_reduction_unit():
    """gs_device.h wave_sum10_transposed: every lane ends with the 64-lane total of the value its lane bits 0, 1, 4, 5 select."""
    import ctypes
    from diff_gaussian_rasterization import _C
    lib = _C.load_library()
    lib.gsr_debug_wave_reduce10.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(0)
    x = rng.normal(size=(64, 10)).astype(np.float32)
    xin = torch.tensor(x, device="cuda")
    out = torch.zeros(64, device="cuda")
    assert lib.gsr_debug_wave_reduce10(xin.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    def slot(l):   # gs_device.h wave_sum10_slot_of_lane
        b0, b1, b4, b5 = l & 1, (l >> 1) & 1, (l >> 4) & 1, (l >> 5) & 1
        return (5 if b5 else 0) if b1 else (6 if b4 else 1) + b0 + 2 * b5
    want = x.astype(np.float64).sum(0)
    got = out.cpu().numpy()
    assert sorted(set(slot(l) for l in range(64))) == list(range(10))
    for l in range(64):
        assert abs(got[l] - want[slot(l)]) < 1e-4, (l, got[l], want[slot(l)])


def test_huge_tile_grid_uses_global_atomic_binning_fallback():
    """More than HIST_LDS_TILES (12288) tiles: the per-block LDS histogram does not fit, binning falls back to one global
    atomic per instance. 2048x1616 -> 128 x 101 = 12928 tiles."""
    cam = make_camera(2048, 1616)
    g = make_gaussians(4000, cam, seed=6, scale_mean=0.02)
    gc, gd = make_cotangents(cam)
    bg = np.zeros(3, np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    oh, gh = hip_run(g, cam, bg, gc, gd)
    _check(compare(oh, gh, oo, go), nt_tol=8)   # 3.3 M pixels


@pytest.mark.gpu
@pytest.mark.parametrize("P", [300_000, 2_200_000])
def test_many_gaussian_blocks_column_scan_variants(P):
    """More than 256 Gaussian blocks of 1024 (the 64-segment variant of the per-tile column scan keeps up to 2048 block rows in
    registers; BASELINE config #5 has 1954) and more than 2048 (its serial fall-back): parity with the oracle on a small image, where
    the oracle stays cheap."""
    cam = make_camera(128, 96)
    g = make_gaussians(P, cam, seed=8, scale_mean=0.002)
    gc, gd = make_cotangents(cam)
    bg = np.array([0.3, 0.3, 0.3], np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    oh, gh = hip_run(g, cam, bg, gc, gd)
    _check(compare(oh, gh, oo, go), nt_tol=max(8, P // 50_000))


@pytest.mark.gpu
def test_speculative_binning_and_its_overflow_redo():
    """The second forward pass of a host thread is enqueued without waiting for R (buffer sized from the previous frame).
    Same scene again: bitwise the same result as the waited-for first pass. A much larger scene next: the speculative capacity
    overflows, the kernels skip, the host redoes them on an exact buffer -- parity with the oracle must hold, as for the small
    scene rendered after it (capacity now far too large) and for a frame with nothing visible in between."""
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    cam = make_camera(160, 128)
    small = make_gaussians(1500, cam, seed=21, sh_degree=1)
    big = make_gaussians(30000, cam, seed=22, sh_degree=1)
    behind = dict(small)
    behind["means3D"] = small["means3D"] * np.array([1, 1, -1], np.float32)      # everything behind the camera: R = 0
    gc, gd = make_cotangents(cam, seed=23)
    first = hip_run(small, cam, bg, gc, gd)
    for scene, tag in ((small, "same scene, speculative"), (big, "overflow -> redo"), (small, "oversized capacity"),
                       (behind, "nothing visible"), (small, "after an empty frame")):
        out_h, gr_h = hip_run(scene, cam, bg, gc, gd)
        out_o, _, gr_o = oracle_run(scene, cam, bg, gc, gd)
        m = compare(out_h, gr_h, out_o, gr_o, tag)
        worst = max(v for k, v in m.items() if isinstance(v, float))
        assert worst < 1e-3 and m["radii_mismatch"] == 0, m
        if scene is small:
            assert np.array_equal(out_h["color"], first[0]["color"]), tag
            for k, v in gr_h.items():
                assert v is None or np.array_equal(v, first[1][k]), (tag, k)


@pytest.mark.gpu
def test_pose_gradient_shapes_follow_the_inputs():
    """theta / rho of shape (3,) (the reference's camera parameters) and of shape (1,3) both receive .grad of their own shape."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    cam = make_camera(64, 48)
    g = make_gaussians(300, cam, seed=5)
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device="cuda", requires_grad=rg)
    rs = GaussianRasterizationSettings(
        image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=T([0, 0, 0]), scale_modifier=1.0,
        viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), projmatrix_raw=T(cam.projmatrix_raw), sh_degree=0,
        campos=T(cam.campos), prefiltered=False, debug=False)
    got = []
    for shape in ((3,), (1, 3)):
        theta, rho = T(np.zeros(shape), True), T(np.zeros(shape), True)
        color, *_ = GaussianRasterizer(rs)(means3D=T(g["means3D"]), means2D=T(np.zeros((300, 3))), opacities=T(g["opacities"]),
                                           shs=T(g["shs"]), scales=T(g["scales"]), rotations=T(g["rotations"]), theta=theta, rho=rho)
        color.sum().backward()
        assert tuple(theta.grad.shape) == shape and tuple(rho.grad.shape) == shape
        got.append((theta.grad.reshape(-1).cpu().numpy(), rho.grad.reshape(-1).cpu().numpy()))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert np.abs(got[0][0]).sum() > 0


ADVERSARIAL = ["needles", "giants", "opaque_wall", "tiny_image", "big_modifier", "far_needles"]


@pytest.mark.parametrize("kind", ADVERSARIAL)
def test_adversarial_scenes_for_the_quadrant_cull_and_the_chunked_backward(kind):
    """Shapes that stress what is NOT in the reference: the exact ellipse-vs-quadrant cull (needles seen end-on and from far
    away, Gaussians larger than the image), early termination + checkpoints (an opaque wall in front of everything, long
    lists), and sizes below one tile."""
    rng = np.random.default_rng(41)
    W, H, P = (200, 136, 3000)
    if kind == "tiny_image":
        W, H, P = 17, 5, 400
    cam = make_camera(W, H)
    g = make_gaussians(P, cam, seed=42, sh_degree=1, scale_mean=0.01)
    smod = 1.0
    if kind in ("needles", "far_needles"):
        s = g["scales"].copy()
        s[:, 0] *= 40.0 if kind == "needles" else 400.0       # 1:40 / 1:400 anisotropy, random orientations
        s[:, 1:] *= 0.3
        g["scales"] = s
        if kind == "far_needles":
            g["opacities"] = np.clip(g["opacities"] * 0 + rng.uniform(0.5, 0.99, g["opacities"].shape), 0, 0.99).astype(np.float32)
    elif kind == "giants":
        g["scales"] = (g["scales"] * rng.choice([1.0, 60.0], size=(P, 1), p=[0.97, 0.03])).astype(np.float32)
    elif kind == "opaque_wall":
        n = P // 3                                              # a dense, nearly opaque layer at the front
        g["means3D"][:n, 2] = 0.6
        g["means3D"][:n, 0] = rng.uniform(-1, 1, n) * 0.6 * cam.tanfovx
        g["means3D"][:n, 1] = rng.uniform(-1, 1, n) * 0.6 * cam.tanfovy
        g["opacities"][:n] = 0.99
        g["scales"][:n] = 0.03
    elif kind == "big_modifier":
        smod = 3.0
    gc, gd = make_cotangents(cam, seed=43)
    bg = np.array([0.3, 0.6, 0.9], np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd, scale_modifier=smod)
    oh, gh = hip_run(g, cam, bg, gc, gd, scale_modifier=smod)
    m = compare(oh, gh, oo, go)
    m["P"] = P
    if kind not in ("needles", "far_needles"):
        _check(m, nt_tol=6)
        return
    # Needles are ill-conditioned in fp32 (the reference's own arithmetic): the fp32 oracle itself is up to 1e-2 .. 1e-1 away
    # from the fp64 oracle on these scenes. The bar here: the HIP path is no further from the fp64 result than the fp32
    # restatement of the reference is (x1.5), and the discrete outputs agree.
    o64, _, g64 = oracle_run(g, cam, bg, gc, gd, scale_modifier=smod, dtype=np.float64)
    m_h64, m_o64 = compare(oh, gh, o64, g64), compare(oo, {k: go[ko] for k, ko in
                                                           (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"),
                                                            ("shs", "dL_dsh"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"))}, o64, g64)
    assert m["radii_mismatch"] == 0 and m["visible_mismatch"] == 0, m
    for k, v in m_h64.items():
        if isinstance(v, float) and k in m_o64:
            assert v <= max(1.5 * m_o64[k], IMG_TOL if not k.startswith("g_") else GRAD_TOL), (k, v, m_o64[k])


def test_randomised_scene_shapes_back_to_back():
    """Random image sizes, Gaussian counts, scales (tile lists from a handful to several thousand entries, across the 1024 / 4096 sort
    thresholds) and SH degrees, rendered back to back in one process so that the speculative binning capacity inherited from the
    previous scene is alternately far too small and far too large."""
    rng = np.random.default_rng(2)
    for it in range(14):
        W, H = int(rng.integers(17, 330)), int(rng.integers(17, 250))
        P = int(rng.choice([1, 7, 300, 3000, 12000, 30000]))
        sm = float(rng.choice([0.002, 0.01, 0.05, 0.2]))
        deg = int(rng.integers(0, 4))
        cam = make_camera(W, H)
        g = make_gaussians(P, cam, seed=int(rng.integers(1 << 30)), sh_degree=deg, scale_mean=sm)
        gc, gd = make_cotangents(cam, seed=it)
        bg = rng.uniform(0, 1, 3).astype(np.float32)
        oo, st, go = oracle_run(g, cam, bg, gc, gd)
        oh, gh = hip_run(g, cam, bg, gc, gd)
        m = compare(oh, gh, oo, go)
        m.update(P=P, case=(W, H, P, sm, deg))
        _check(m, nt_tol=4)


@pytest.mark.gpu
def test_large_gaussians_many_instances_per_gaussian():
    """SLAM-shaped map: every Gaussian covers tens of tiles, so geometry_bwd takes its wave-cooperative gather (more than 8 instance
    slots per Gaussian in a block) and the tile lists run into the thousands; parity with the oracle as everywhere else."""
    cam = make_camera(208, 160)
    g = make_gaussians(2600, cam, seed=17, sh_degree=1, scale_mean=0.06)
    g["opacities"][:] *= 0.25
    gc, gd = make_cotangents(cam, seed=18)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    oo, st, go = oracle_run(g, cam, bg, gc, gd)
    assert st.num_rendered > 12 * 2600                          # > 8 slots per Gaussian on average: the heavy path
    oh, gh = hip_run(g, cam, bg, gc, gd)
    _check(compare(oh, gh, oo, go))
    oh2, gh2 = hip_run(g, cam, bg, gc, gd)                      # bit-reproducible
    for k, v in gh.items():
        assert v is None or np.array_equal(v, gh2[k]), k
