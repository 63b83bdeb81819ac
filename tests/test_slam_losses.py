"""SURVEY.md 8f rank 2 -- fused mapping loss. Golden vectors come from the reference's own utils/slam_utils.py:get_loss_mapping
(tests/golden/make_golden_loss.py). CPU: this repo's mask/weight logic + the plain-torch loss expression (oracle/loss_oracle.py)
reproduce them. GPU: slam_losses.get_loss_mapping (two HIP kernels through the C ABI) reproduces value and gradients."""
import os
import types

import numpy as np
import pytest
import torch

import util  # noqa: F401  (puts the repo and the package on sys.path)

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_loss.npz"), allow_pickle=False)
CASES = [str(c) for c in FX["cases"]]
CONFIG = {"Training": {"monocular": False, "rgb_boundary_threshold": float(FX["cfg_thr"]), "alpha": float(FX["cfg_alpha"])}}


def _viewpoint(dev):
    return types.SimpleNamespace(
        original_image=torch.tensor(FX["gt_image"], device=dev), depth=FX["gt_depth"],
        exposure_a=torch.nn.Parameter(torch.tensor([0.07], device=dev)), exposure_b=torch.nn.Parameter(torch.tensor([-0.03], device=dev)),
        motion_mask=torch.tensor(FX["motion_mask"], device=dev), uid=3)


def _kwargs(name, dev):
    init, rm_dyn, dyn, use_mask, has_alpha = [bool(v) for v in FX[f"{name}/flags"]]
    kw = dict(initialization=init, rm_dynamic=rm_dyn, dynamic=dyn, mask=torch.tensor(FX["mask"], device=dev) if use_mask else None)
    if has_alpha:
        kw["alpha"] = float(FX[f"{name}/alpha"])
    return kw


def _check(name, loss, image, depth, vp, tol):
    loss.backward()
    assert abs(float(loss) - float(FX[f"{name}/loss"])) < tol * max(1.0, abs(float(FX[f"{name}/loss"])))
    assert util.rel_l1(image.grad.cpu().numpy(), FX[f"{name}/g_image"]) < tol
    assert util.rel_l1(depth.grad.cpu().numpy(), FX[f"{name}/g_depth"]) < tol
    if FX[f"{name}/g_a"].size:
        assert abs(float(vp.exposure_a.grad) - float(FX[f"{name}/g_a"][0])) < 10 * tol * max(1e-3, abs(float(FX[f"{name}/g_a"][0])))
        assert abs(float(vp.exposure_b.grad) - float(FX[f"{name}/g_b"][0])) < 10 * tol * max(1e-3, abs(float(FX[f"{name}/g_b"][0])))
    else:
        assert vp.exposure_a.grad is None and vp.exposure_b.grad is None


@pytest.mark.parametrize("name", CASES)
def test_weight_logic_and_loss_expression_reproduce_the_reference_on_cpu(name):
    from slam_losses import mapping_loss_weights
    from oracle.loss_oracle import weighted_l1_loss_reference

    vp = _viewpoint("cpu")
    kw = _kwargs(name, "cpu")
    image = torch.tensor(FX[f"{name}/image"], requires_grad=True)
    depth = torch.tensor(FX[f"{name}/depth"], requires_grad=True)
    gt_depth = torch.tensor(FX["gt_depth"])[None]
    w_rgb, w_dep = mapping_loss_weights(CONFIG, vp, vp.original_image, gt_depth, kw["rm_dynamic"], kw["mask"], kw["dynamic"])
    a, b = (None, None) if kw["initialization"] else (vp.exposure_a, vp.exposure_b)
    loss = weighted_l1_loss_reference(image, depth, vp.original_image, gt_depth, w_rgb, w_dep, a, b, kw.get("alpha", CONFIG["Training"]["alpha"]))
    _check(name, loss, image, depth, vp, 1e-5)


def test_product_loss_refuses_cpu_tensors():
    from slam_losses import get_loss_mapping

    vp = _viewpoint("cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_loss_mapping(CONFIG, torch.tensor(FX["plain/image"]), torch.tensor(FX["plain/depth"]), vp, None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_fused_mapping_loss_matches_the_reference(name):
    from slam_losses import get_loss_mapping

    vp = _viewpoint("cuda")
    image = torch.tensor(FX[f"{name}/image"], device="cuda", requires_grad=True)
    depth = torch.tensor(FX[f"{name}/depth"], device="cuda", requires_grad=True)
    loss = get_loss_mapping(CONFIG, image, depth, vp, None, **_kwargs(name, "cuda"))
    _check(name, loss, image, depth, vp, 2e-5)


@pytest.mark.gpu
def test_fused_loss_scales_with_the_upstream_gradient_and_is_reproducible():
    from slam_losses import get_loss_mapping

    grads = []
    for scale in (1.0, 1.0, 3.0):
        vp = _viewpoint("cuda")
        image = torch.tensor(FX["plain/image"], device="cuda", requires_grad=True)
        depth = torch.tensor(FX["plain/depth"], device="cuda", requires_grad=True)
        (get_loss_mapping(CONFIG, image, depth, vp, None) * scale).backward()
        grads.append((image.grad.clone(), depth.grad.clone(), vp.exposure_a.grad.clone()))
    assert all(torch.equal(x, y) for x, y in zip(grads[0], grads[1]))                 # bit-reproducible
    assert all(torch.allclose(3.0 * x, y, rtol=1e-6, atol=0) for x, y in zip(grads[0], grads[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("terms,H,W", [(2, 48, 64), (1, 37, 53), (4, 480, 640)])
def test_masked_l1_equals_the_flow_terms_expression(terms, H, W):
    """The optical-flow terms as the reference writes them (utils/slam_backend.py:486-488,503-505):
    flow_weight * |flow * ~mask[..., None] - render[:2].permute(1, 2, 0) * ~mask[..., None]|.mean(), summed over the directions."""
    from slam_losses import masked_l1

    g = torch.Generator(device="cpu").manual_seed(terms * H)
    weight = 0.37
    renders = [torch.randn(3, H, W, generator=g).cuda().requires_grad_(True) for _ in range(terms)]
    flows = [torch.randn(H, W, 2, generator=g).cuda() for _ in range(terms)]
    moving = [(torch.rand(H, W, generator=g) < 0.3).cuda() for _ in range(terms)]
    want = 0.0
    for r, f, m in zip(renders, flows, moving):
        keep = (~m)[..., None]
        want = want + weight * torch.abs(f * keep - r[:2].permute(1, 2, 0) * keep).mean()
    (want * 2.0).backward()
    want_grads = [r.grad.clone() for r in renders]
    for r in renders:
        r.grad = None
    packed = []
    for r, f, m in zip(renders, flows, moving):
        mk = (~m).to(torch.float32)[None]
        packed.append((r, (f.permute(2, 0, 1) * mk).contiguous(), mk))
    got = masked_l1(weight, packed, channels=2)
    (got * 2.0).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    for r, wg in zip(renders, want_grads):
        assert torch.allclose(r.grad, wg, rtol=1e-6, atol=1e-12) and float(r.grad[2].abs().max()) == 0.0
    again = masked_l1(weight, packed, channels=2)
    assert torch.equal(again, got)                                                     # fixed summation order
    with pytest.raises(RuntimeError):
        masked_l1(weight, [(renders[0], packed[0][1][:1], packed[0][2])], channels=2)


TCASES = [str(c) for c in FX["tracking_cases"]]


def _tracking_viewpoint(dev):
    vp = _viewpoint(dev)
    vp.grad_mask = torch.tensor(FX["grad_mask"], device=dev)
    return vp


@pytest.mark.parametrize("name", TCASES)
def test_tracking_weights_and_expression_reproduce_the_reference_on_cpu(name):
    from slam_losses import tracking_loss_weights
    from oracle.loss_oracle import weighted_l1_loss_reference

    vp = _tracking_viewpoint("cpu")
    rm_dyn, use_mask = [bool(v) for v in FX[f"{name}/flags"]]
    image = torch.tensor(FX[f"{name}/image"], requires_grad=True)
    depth = torch.tensor(FX[f"{name}/depth"], requires_grad=True)
    gt_depth = torch.tensor(FX["gt_depth"])[None]
    w_rgb, w_dep = tracking_loss_weights(CONFIG, vp, vp.original_image, gt_depth, rm_dyn, torch.tensor(FX["mask"]) if use_mask else None)
    loss = weighted_l1_loss_reference(image, depth, vp.original_image, gt_depth, w_rgb, w_dep, vp.exposure_a, vp.exposure_b,
                                      CONFIG["Training"]["alpha"], opacity=torch.tensor(FX[f"{name}/opacity"]))
    _check(name, loss, image, depth, vp, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TCASES)
def test_fused_tracking_loss_matches_the_reference(name):
    from slam_losses import get_loss_tracking

    vp = _tracking_viewpoint("cuda")
    rm_dyn, use_mask = [bool(v) for v in FX[f"{name}/flags"]]
    image = torch.tensor(FX[f"{name}/image"], device="cuda", requires_grad=True)
    depth = torch.tensor(FX[f"{name}/depth"], device="cuda", requires_grad=True)
    opacity = torch.tensor(FX[f"{name}/opacity"], device="cuda")
    loss = get_loss_tracking(CONFIG, image, depth, opacity, vp, rm_dynamic=rm_dyn, mask=torch.tensor(FX["mask"], device="cuda") if use_mask else None)
    _check(name, loss, image, depth, vp, 2e-5)


SCASES = [str(c) for c in FX["ssim_cases"]]


def _ssim_case(name, dev):
    img1 = torch.tensor(FX[f"{name}/img1"], device=dev, requires_grad=True)
    img2 = torch.tensor(FX[f"{name}/img2"], device=dev)
    mk = torch.tensor(FX[f"{name}/mask"], device=dev) if FX[f"{name}/mask"].size else None
    return img1, img2, mk


@pytest.mark.parametrize("name", SCASES)
def test_ssim_restatement_reproduces_the_reference_on_cpu(name):
    from oracle.loss_oracle import ssim_reference

    img1, img2, mk = _ssim_case(name, "cpu")
    v = ssim_reference(img1, img2, mk)
    v.backward()
    assert abs(float(v) - float(FX[f"{name}/value"])) < 1e-6
    assert util.rel_l1(img1.grad.numpy(), FX[f"{name}/g_img1"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCASES)
def test_fused_ssim_matches_the_reference(name):
    from slam_losses import ssim

    img1, img2, mk = _ssim_case(name, "cuda")
    v = ssim(img1, img2, mask=mk)
    (0.2 * (1.0 - v)).backward()                       # the way the losses use it: lambda_dssim * (1 - ssim)
    assert abs(float(v) - float(FX[f"{name}/value"])) < 2e-6
    assert util.rel_l1(img1.grad.cpu().numpy(), -0.2 * FX[f"{name}/g_img1"]) < 2e-5


@pytest.mark.gpu
def test_fused_densification_stats_match_the_reference_expressions():
    import types
    from slam_losses import add_densification_stats

    gen = torch.Generator().manual_seed(5)
    P = 5000
    radii = torch.where(torch.rand(P, generator=gen) < 0.3, 0, torch.randint(1, 40, (P,), generator=gen)).to(torch.int32).cuda()
    vs = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vs.grad = torch.randn(P, 3, generator=gen).cuda()
    mk = lambda *s: torch.rand(*s, generator=gen).cuda() * 10
    g = types.SimpleNamespace(max_radii2D=mk(P), xyz_gradient_accum=mk(P, 1), denom=mk(P, 1).round())
    want = types.SimpleNamespace(max_radii2D=g.max_radii2D.clone(), xyz_gradient_accum=g.xyz_gradient_accum.clone(), denom=g.denom.clone())
    vis = radii > 0                                                                  # utils/slam_backend.py:712-720, gaussian_model.py:973-977
    want.max_radii2D[vis] = torch.max(want.max_radii2D[vis], radii[vis])
    want.xyz_gradient_accum[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)
    want.denom[vis] += 1
    add_densification_stats(g, vs, radii)
    assert torch.equal(g.max_radii2D, want.max_radii2D) and torch.equal(g.denom, want.denom)
    assert torch.allclose(g.xyz_gradient_accum, want.xyz_gradient_accum, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("compute_value", [True, False])
def test_cpp_loss_node_equals_the_python_node_bitwise(compute_value):
    """weighted_l1_loss's autograd node in C++ (torch_glue.cpp WeightedL1Node) against the Python node: same value, same gradients, with
    and without exposure parameters / opacity / weights, and in the back-propagate-only mode."""
    import slam_losses
    from diff_gaussian_rasterization import _C
    if _C._glue is None or not hasattr(_C._glue, "weighted_l1_autograd"):
        pytest.skip("native glue not built")
    g = torch.Generator(device="cpu").manual_seed(3)
    H, W = 60, 84
    R = lambda *s: torch.rand(*s, generator=g).cuda()

    def run(native, variant):
        slam_losses._NATIVE_NODE = native
        image, depth = R(3, H, W).requires_grad_(True), R(1, H, W).requires_grad_(True)
        kw = {}
        if variant >= 1:
            kw.update(exposure_a=torch.tensor([0.05], device="cuda", requires_grad=True), exposure_b=torch.tensor([-0.02], device="cuda", requires_grad=True))
        if variant >= 2:
            kw.update(w_rgb=(R(1, H, W) > 0.3).float(), w_depth=(R(1, H, W) > 0.5).float() * 2, opacity=R(1, H, W))
        loss = slam_losses.weighted_l1_loss(image, depth, R(3, H, W), R(1, H, W), alpha=0.9, compute_value=compute_value, **kw)
        (loss * 1.7).backward()
        return [loss.detach(), image.grad, depth.grad] + [kw[k].grad for k in ("exposure_a", "exposure_b") if k in kw]

    try:
        for variant in (0, 1, 2):
            g.manual_seed(3 + variant)
            a = run(True, variant)
            g.manual_seed(3 + variant)
            b = run(False, variant)
            assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b)), variant
    finally:
        slam_losses._NATIVE_NODE = True
