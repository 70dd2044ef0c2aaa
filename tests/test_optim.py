"""FlatOptimizer: the reference's optimizer interface over flat buffers + one fused update launch.

Oracle: torch.optim.{SGD, Adam, AdamW} on the CPU -- the reference's own optimizer objects (torch_src/session_helper.py:48-53);
the GPU tests compare the libfgcn update with them step by step on identical parameters and gradients (tolerance: 2e-6
relative per step sequence, float32 elementwise arithmetic in the same operation order)."""
import copy

import pytest
import torch
import torch.nn as nn


def small_model(seed=0):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Conv2d(3, 5, (9, 1)), nn.BatchNorm2d(5), nn.Conv2d(5, 7, 1), nn.Linear(7, 3))


CASES = [("SGD", dict()), ("SGD", dict(momentum=0.9, weight_decay=1e-4, nesterov=True)),
         ("SGD", dict(momentum=0.8, dampening=0.1)), ("ADAM", dict(weight_decay=0.01)), ("ADAM", dict(betas=(0.8, 0.99), eps=1e-6)),
         ("ADAMW", dict()), ("ADAMW", dict(weight_decay=0.1))]
TORCH = {"SGD": torch.optim.SGD, "ADAM": torch.optim.Adam, "ADAMW": torch.optim.AdamW}


def test_flat_optimizer_rehomes_parameters_and_keeps_the_torch_interface():
    from fusion_gcn_amd.optim import FlatOptimizer, create_optimizer
    m = small_model()
    before = [p.detach().clone() for p in m.parameters()]
    opt = create_optimizer("adam", m, 0.1, weight_decay=0.01)
    assert isinstance(opt, torch.optim.Optimizer) and isinstance(opt, FlatOptimizer)
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
        assert p.data_ptr() >= opt.flat.data_ptr() and p.data_ptr() % 16 == 0          # lives in the flat buffer, aligned
    assert opt.flat.numel() % 4 == 0 and opt.flat.numel() == opt.grads.flat.numel()
    # torch's schedulers drive it through param_groups, as create_learning_rate_scheduler does in the reference
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=20)
    sched.step()
    assert 0 < opt.param_groups[0]["lr"] < 0.1
    ref = torch.optim.Adam(small_model().parameters(), 0.1, weight_decay=0.01)
    mine, theirs = opt.state_dict(), ref.state_dict()
    assert mine["state"] == {} and theirs["state"] == {}
    assert mine["param_groups"][0]["params"] == theirs["param_groups"][0]["params"]
    for k in ("lr", "betas", "eps", "weight_decay"):
        assert k in mine["param_groups"][0]
    with pytest.raises(ValueError):
        FlatOptimizer(small_model().parameters(), "RMSPROP", 0.1)
    with pytest.raises(NotImplementedError):
        FlatOptimizer(small_model().parameters(), "ADAM", 0.1, amsgrad=True)
    with pytest.raises(TypeError):
        FlatOptimizer(small_model().parameters(), "SGD", 0.1, betas=(0.9, 0.99))


def test_step_fails_loudly_without_a_gpu():
    from fusion_gcn_amd import _lib
    from fusion_gcn_amd.optim import FlatOptimizer
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    m = small_model()
    opt = FlatOptimizer(m.parameters(), "SGD", 0.1)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    with pytest.raises(_lib.FgcnError):
        opt.step()


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", CASES)
def test_fused_update_matches_torch_optim(name, args):
    from fusion_gcn_amd.optim import FlatOptimizer
    dev = torch.device("cuda:0")
    ref_model = small_model(3)
    model = copy.deepcopy(ref_model).to(dev)
    ref = TORCH[name](ref_model.parameters(), 0.05, **args)
    opt = FlatOptimizer(model.parameters(), name, 0.05, **args)
    sched_r = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref, T_0=3)
    sched_o = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=3)
    g = torch.Generator().manual_seed(7)
    for it in range(6):
        ref.zero_grad()
        opt.zero_grad()
        for pr, po in zip(ref_model.parameters(), model.parameters()):
            grad = torch.randn(pr.shape, generator=g) * (1.0 + it)
            pr.grad = grad.clone()
            po.grad = grad.to(dev)
        ref.step()
        opt.step()
        sched_r.step()
        sched_o.step()
        assert abs(ref.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12
        for pr, po in zip(ref_model.parameters(), model.parameters()):
            err = float((po.detach().cpu() - pr.detach()).norm() / pr.detach().norm())
            assert err < 2e-6, (name, args, it, err)
    # the raw-pointer update is visible to autograd's version counters (the blocks key their packed-weight cache on them)
    v0 = [p._version for p in model.parameters()]
    for po in model.parameters():
        po.grad = torch.zeros_like(po)
    opt.step()
    assert all(p._version > v for p, v in zip(model.parameters(), v0))
    # padding between the views stays zero (the kernel runs over the whole buffer)
    used = torch.zeros_like(opt.flat, dtype=torch.bool)
    for v, p in zip(opt.grads.views, opt.params):
        used[v.storage_offset():v.storage_offset() + p.numel()] = True
    assert float(opt.flat[~used].abs().sum()) == 0.0
    # the state dict loads into the torch optimizer of that name and back
    sd = opt.state_dict()
    other = TORCH[name](copy.deepcopy(ref_model).parameters(), 0.05, **args)
    other.load_state_dict(sd)
    opt2 = FlatOptimizer(copy.deepcopy(ref_model).to(dev).parameters(), name, 0.05, **args)
    opt2.load_state_dict(ref.state_dict())
    if name != "SGD":
        assert opt2.steps == 6 and opt.steps == 7
        a, b = opt2.state_dict()["state"][0]["exp_avg_sq"].cpu(), ref.state_dict()["state"][0]["exp_avg_sq"]
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_grad_scale_is_the_data_parallel_average():
    from fusion_gcn_amd.optim import FlatOptimizer
    dev = torch.device("cuda:0")
    a, b = small_model(5).to(dev), small_model(5).to(dev)
    oa, ob = FlatOptimizer(a.parameters(), "ADAM", 0.01, weight_decay=0.01), FlatOptimizer(b.parameters(), "ADAM", 0.01, weight_decay=0.01)
    ob.grad_scale = 0.25
    g = torch.Generator().manual_seed(1)
    for pa, pb in zip(a.parameters(), b.parameters()):
        grad = torch.randn(pa.shape, generator=g).to(dev)
        pa.grad, pb.grad = grad.clone(), grad * 4.0
    oa.step(), ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)


def test_step_refuses_parameters_that_left_the_flat_buffer():
    """A model moved / cast / re-assigned after the optimizer was built would keep training an orphaned copy: step() checks the
    aliasing (pointer comparisons) and raises instead.  Frozen parameters keep torch's state-dict slot numbering."""
    from fusion_gcn_amd._lib import FgcnError
    from fusion_gcn_amd.optim import FlatOptimizer
    m = small_model()
    list(m.parameters())[1].requires_grad_(False)                 # a frozen parameter in the middle
    opt = FlatOptimizer(m.parameters(), "SGD", 0.1, momentum=0.9)
    assert opt._slots() == [0] + list(range(2, len(list(m.parameters()))))
    assert opt.state_dict()["param_groups"][0]["params"] == list(range(len(list(m.parameters()))))
    opt._check_homes()
    p = next(m.parameters())
    p.data = p.data.clone()
    with pytest.raises(FgcnError, match="no longer aliases"):
        opt.step()


def test_unused_parameter_is_an_error_not_a_silent_zero_gradient():
    from fusion_gcn_amd.dp import FlatGradients
    m = small_model()
    g = FlatGradients(m.parameters())
    for p in list(m.parameters())[:-1]:
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match="no gradient"):
        g.gather()
    FlatGradients(m.parameters(), allow_unused=True).gather()


def test_zero_grad_in_place_before_any_backward():
    """opt.zero_grad(set_to_none=False) with no gradient yet (before the first backward, or after a set_to_none zero) points every
    p.grad at its zeroed view of the flat buffer instead of raising 'received no gradient'; allow_unused reaches the buffer."""
    from fusion_gcn_amd.optim import FlatOptimizer
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt = FlatOptimizer(m.parameters(), "SGD", 0.1, allow_unused=True)
    assert opt.grads.allow_unused
    opt.zero_grad(set_to_none=False)
    for p, v in zip(opt.grads.params, opt.grads.views):
        assert p.grad is not None and p.grad.data_ptr() == v.data_ptr() and float(p.grad.abs().sum()) == 0.0
    m(torch.randn(5, 4)).sum().backward()
    assert float(opt.grads.flat.abs().sum()) > 0          # backward accumulated into the flat buffer through the views
    opt.zero_grad()                                          # set_to_none=True
    assert all(p.grad is None for p in m.parameters())
    opt.zero_grad(set_to_none=False)
    assert float(opt.grads.flat.abs().sum()) == 0.0
