"""SURVEY.md 8f rank 2 (second half): FusedAdam must follow torch.optim.Adam step for step on the reference's six parameter groups,
through a learning-rate change, a parameter without gradient, and the densification-style state surgery the reference performs."""
import pytest
import torch

import util  # noqa: F401


def _groups(dev, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *shape: torch.nn.Parameter(torch.randn(*shape, generator=g).to(dev))
    return [{"params": [mk(500, 3)], "lr": 1.6e-4, "name": "xyz"}, {"params": [mk(500, 1, 3)], "lr": 2.5e-3, "name": "f_dc"},
            {"params": [mk(500, 15, 3)], "lr": 1.25e-4, "name": "f_rest"}, {"params": [mk(500, 1)], "lr": 0.05, "name": "opacity"},
            {"params": [mk(500, 3)], "lr": 0.001, "name": "scaling"}, {"params": [mk(500, 4)], "lr": 0.001, "name": "rotation"}]


@pytest.mark.gpu
def test_fused_adam_tracks_torch_adam():
    from fused_adam import FusedAdam

    ref_groups, fused_groups = _groups("cpu", 3), _groups("cuda", 3)
    ref = torch.optim.Adam(ref_groups, lr=0.0, eps=1e-15)
    opt = FusedAdam(fused_groups, lr=0.0, eps=1e-15)
    gen = torch.Generator().manual_seed(4)
    for it in range(6):
        for gr, gf in zip(ref.param_groups, opt.param_groups):
            pr, pf = gr["params"][0], gf["params"][0]
            if gr["name"] == "f_rest" and it == 2:          # a tensor that received no gradient this iteration
                pr.grad = pf.grad = None
                continue
            g = torch.randn(pr.shape, generator=gen) * (10.0 ** (-it))
            pr.grad, pf.grad = g.clone(), g.to("cuda")
        if it == 3:                                          # update_learning_rate (gaussian_model.py:492-505)
            ref.param_groups[0]["lr"] = opt.param_groups[0]["lr"] = 5e-5
        ref.step()
        opt.step()
        for gr, gf in zip(ref.param_groups, opt.param_groups):
            pr, pf = gr["params"][0], gf["params"][0]
            assert torch.allclose(pf.detach().cpu(), pr.detach(), rtol=2e-6, atol=1e-7), (it, gr["name"])
            if pr in ref.state:
                assert torch.allclose(opt.state[pf]["exp_avg"].cpu(), ref.state[pr]["exp_avg"], rtol=2e-6, atol=1e-12)
                assert torch.allclose(opt.state[pf]["exp_avg_sq"].cpu(), ref.state[pr]["exp_avg_sq"], rtol=2e-6, atol=1e-20)
    # densification-style surgery: extend a parameter and its moments (cat_tensors_to_optimizer, gaussian_model.py:792-824)
    group = opt.param_groups[0]
    old = group["params"][0]
    st = opt.state.pop(old)
    st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros(10, 3, device="cuda")))
    st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros(10, 3, device="cuda")))
    new = torch.nn.Parameter(torch.cat((old.detach(), torch.ones(10, 3, device="cuda"))))
    group["params"][0] = new
    opt.state[new] = st
    new.grad = torch.ones_like(new)
    before = new.detach().clone()
    opt.step()
    assert not torch.equal(new.detach(), before) and torch.isfinite(new).all()


def test_fused_adam_uses_torch_for_what_it_does_not_fuse():
    from fused_adam import FusedAdam

    p = torch.nn.Parameter(torch.ones(4))                    # CPU parameter: torch.optim.Adam's own step()
    opt = FusedAdam([p], lr=0.1)
    p.grad = torch.ones(4)
    opt.step()
    assert torch.allclose(p.detach(), torch.full((4,), 0.9))


@pytest.mark.gpu
def test_device_count_adam_tracks_torch_adam_and_replays_in_a_graph():
    """fused_adam.DeviceCountAdam (the node network's optimizer: step counts on the device, two launches per step) against torch.optim.Adam on
    27 tensors in two groups: step for step, through a tensor without gradient, state surgery in torch's layout, and as a captured graph whose
    replays advance the counts (the bias corrections of step t, not of the captured step)."""
    from fused_adam import DeviceCountAdam

    def groups(dev):
        g = torch.Generator().manual_seed(9)
        mk = lambda *shape: torch.nn.Parameter(torch.randn(*shape, generator=g).to(dev))
        mlp = [mk(256, 84), mk(256)] + [t for _ in range(7) for t in (mk(256, 256), mk(256))] + [t for c in (3, 3, 4, 4) for t in (mk(c, 256), mk(c))]
        return [{"params": mlp, "lr": 8e-4, "name": "mlp"}, {"params": [mk(512, 3), mk(512), mk(512)], "lr": 8e-4, "name": "nodes"}]

    ref = torch.optim.Adam(groups("cpu"), lr=0.0, eps=1e-15)
    opt = DeviceCountAdam(groups("cuda"), lr=0.0, eps=1e-15)
    pairs = [(pr, pf) for gr, gf in zip(ref.param_groups, opt.param_groups) for pr, pf in zip(gr["params"], gf["params"])]
    assert len(pairs) == 27
    gen = torch.Generator().manual_seed(10)

    def grads(it, skip=None):
        for k, (pr, pf) in enumerate(pairs):
            if k == skip:
                pr.grad = pf.grad = None
                continue
            g = torch.randn(pr.shape, generator=gen) * (10.0 ** (-(it % 4)))
            pr.grad = g.clone()
            if pf.grad is None:
                pf.grad = g.to("cuda")
            else:
                pf.grad.copy_(g)                     # (in place: a captured step reads these addresses)

    def check(it):
        for k, (pr, pf) in enumerate(pairs):
            assert torch.allclose(pf.detach().cpu(), pr.detach(), rtol=3e-6, atol=1e-7), (it, k)
            if pr in ref.state:
                assert torch.allclose(opt.state[pf]["exp_avg_sq"].cpu(), ref.state[pr]["exp_avg_sq"], rtol=3e-6, atol=1e-30)
                assert int(opt.state[pf]["step"].item()) == int(ref.state[pr]["step"].item())

    for it in range(5):                              # step 0 creates the state (torch's own step), the others are this library's launches
        grads(it, skip=3 if it == 2 else None)
        ref.step()
        opt.step()
        check(it)
    assert opt._coefficients is not None             # (the fused path ran)
    # state surgery as extend_node_from_point does it: a parameter replaced by a longer one, moments extended with zeros, the count kept
    gr, gf = ref.param_groups[1], opt.param_groups[1]
    for o, grp, dev in ((ref, gr, "cpu"), (opt, gf, "cuda")):
        old = grp["params"][0]
        st = o.state.pop(old)
        p = torch.nn.Parameter(torch.cat((old.detach(), torch.ones((4, 3), device=dev)), 0))
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros((4, 3), device=dev)), 0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros((4, 3), device=dev)), 0)
        o.state[p] = st
        grp["params"][0] = p
    pairs[24] = (gr["params"][0], gf["params"][0])
    grads(5)
    ref.step()
    opt.step()
    check(5)
    # captured once, replayed three times with new gradients
    grads(6)
    ref.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        graph.capture_begin()
        opt.step()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    graph.replay()
    check(6)
    for it in (7, 8):
        grads(it)
        ref.step()
        graph.replay()
        check(it)
