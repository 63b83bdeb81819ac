"""Analytic known-answer tests that pin the oracle (oracle/gs_oracle.c) independently of any other implementation
(the reference has no golden vectors of its own)."""
import numpy as np
import pytest

from util import oracle, make_camera


def _one(cam, mean, scale, opacity, rgb, bg=(0.2, 0.4, 0.6), dtype=np.float32):
    sh = ((np.asarray(rgb, np.float64) - 0.5) / 0.28209479177387814).reshape(1, 1, 3)
    return oracle.rasterize_forward(
        bg=np.asarray(bg), means3D=np.asarray([mean]), opacities=np.asarray([[opacity]]), shs=sh,
        scales=np.asarray([[scale] * 3]), rotations=np.asarray([[1.0, 0, 0, 0]]), viewmatrix=cam.viewmatrix,
        projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=cam.H,
        image_width=cam.W, sh_degree=0, dtype=dtype)


def _cam_centered(W=64, H=48):
    # principal point chosen so that the optical axis lands exactly on pixel (W/2 - 0.5 + 0.5, ...) = integer pixel centre
    # ndc2Pix(0, W) = (W - 1) / 2, so shift cx to make a point ON an integer pixel: pick the point instead (see below)
    return make_camera(W, H, fx=60.0, fy=60.0, cx=W / 2, cy=H / 2)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_single_gaussian_on_pixel_centre(dtype):
    cam = _cam_centered()
    z = 2.0
    # pixel (px,py) <-> ndc = (2*px+1)/W - 1 ; x = ndc * tanfov * z
    px, py = 40, 20
    x = ((2 * px + 1) / cam.W - 1) * cam.tanfovx * z
    y = ((2 * py + 1) / cam.H - 1) * cam.tanfovy * z
    o, rgb, bg = 0.7, (0.9, 0.3, 0.1), (0.2, 0.4, 0.6)
    out, st = _one(cam, (x, y, z), 0.05, o, rgb, bg, dtype)
    tol = 2e-5 if dtype == np.float32 else 1e-7   # camera matrices are float32 in both modes
    c = out["color"][:, py, px]
    np.testing.assert_allclose(c, o * np.array(rgb) + (1 - o) * np.array(bg), rtol=tol, atol=tol)   # forward.cu:387
    np.testing.assert_allclose(out["depth"][0, py, px], o * z, rtol=tol)                              # :367,389 (un-normalised)
    np.testing.assert_allclose(out["opacity"][0, py, px], o, rtol=tol)                                # :390
    assert out["radii"][0] > 0 and out["n_touched"][0] >= 1 and out["num_rendered"] >= 1
    # far from the Gaussian only the background remains (:387 with T = 1)
    np.testing.assert_allclose(out["color"][:, 2, 2], bg, rtol=1e-6)
    assert out["opacity"][0, 2, 2] == 0


def test_alpha_clamp_and_min_alpha():
    cam = _cam_centered()
    z, px, py = 2.0, 10, 10
    x = ((2 * px + 1) / cam.W - 1) * cam.tanfovx * z
    y = ((2 * py + 1) / cam.H - 1) * cam.tanfovy * z
    out, _ = _one(cam, (x, y, z), 0.05, 1.0, (1, 1, 1), (0, 0, 0))
    assert abs(out["opacity"][0, py, px] - 0.99) < 1e-6            # alpha = min(0.99, .), forward.cu:353
    out, _ = _one(cam, (x, y, z), 0.05, 1.0 / 300.0, (1, 1, 1), (0, 0, 0))
    assert out["opacity"].max() == 0.0                              # alpha < 1/255 is skipped everywhere, :354
    assert out["n_touched"][0] == 0 and out["radii"][0] > 0


def test_near_cull_and_ordering():
    cam = _cam_centered()
    out, _ = _one(cam, (0.0, 0.0, 0.2), 0.05, 0.9, (1, 0, 0))      # z <= 0.2 culled, auxiliary.h:154
    assert out["radii"][0] == 0 and out["num_rendered"] == 0
    out, _ = _one(cam, (0.0, 0.0, 0.2001), 0.001, 0.9, (1, 0, 0))
    assert out["radii"][0] > 0
    # two stacked Gaussians: front one dominates regardless of input order
    z1, z2, px, py = 1.0, 3.0, 32, 24
    def xy(z):
        return ((2 * px + 1) / cam.W - 1) * cam.tanfovx * z, ((2 * py + 1) / cam.H - 1) * cam.tanfovy * z
    (x1, y1), (x2, y2) = xy(z1), xy(z2)
    C0 = 0.28209479177387814
    sh = ((np.array([[1.0, 0, 0], [0, 0, 1.0]]) - 0.5) / C0).reshape(2, 1, 3)
    def run(order):
        m = np.array([[x1, y1, z1], [x2, y2, z2]])[order]
        o, _ = oracle.rasterize_forward(bg=np.zeros(3), means3D=m, opacities=np.array([[0.6], [0.6]]), shs=sh[order],
                                        scales=np.full((2, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (2, 1)),
                                        viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx,
                                        tanfovy=cam.tanfovy, image_height=cam.H, image_width=cam.W, sh_degree=0)
        return o["color"][:, py, px]
    a, b = run([0, 1]), run([1, 0])
    np.testing.assert_allclose(a, b, rtol=1e-6)
    np.testing.assert_allclose(a, [0.6, 0, 0.4 * 0.6], rtol=1e-4)   # C = c1 a1 + c2 a2 (1 - a1)


def test_depth_tie_is_broken_by_index():
    """Stable (tile|depth) sort: equal depths keep Gaussian-index order (rasterizer_impl.cu:98-108,306-311)."""
    cam = _cam_centered()
    P = 6
    m = np.tile([[0.0, 0.0, 2.0]], (P, 1))
    out, st = oracle.rasterize_forward(bg=np.zeros(3), means3D=m, opacities=np.full((P, 1), 0.5), shs=np.zeros((P, 1, 3)),
                                       scales=np.full((P, 3), 0.01), rotations=np.tile([1.0, 0, 0, 0], (P, 1)),
                                       viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx,
                                       tanfovy=cam.tanfovy, image_height=cam.H, image_width=cam.W, sh_degree=0)
    s = st.state()
    for a, b in s["ranges"]:
        if b > a:
            assert list(s["point_list"][a:b]) == sorted(s["point_list"][a:b])


def test_empty_and_prefiltered():
    cam = _cam_centered()
    out, st = oracle.rasterize_forward(bg=np.ones(3), means3D=np.zeros((0, 3)), opacities=np.zeros((0, 1)), shs=np.zeros((0, 1, 3)),
                                       scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)), viewmatrix=cam.viewmatrix,
                                       projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                       image_height=cam.H, image_width=cam.W)
    assert st is None and out["color"].max() == 0.0                 # rasterize_points.cu:85: outputs stay zero, not bg
    with pytest.raises(RuntimeError):
        oracle.rasterize_forward(bg=np.ones(3), means3D=np.array([[0, 0, 0.1]]), opacities=np.ones((1, 1)), shs=np.zeros((1, 1, 3)),
                                 scales=np.ones((1, 3)), rotations=np.array([[1.0, 0, 0, 0]]), viewmatrix=cam.viewmatrix,
                                 projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                 image_height=cam.H, image_width=cam.W, prefiltered=True)
