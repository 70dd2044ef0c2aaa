"""Host logic of the training-harness mirror (fusion_gcn_amd/session/: reference torch_src/session/procedures/step.py,
batch_train.py, session/session.py:161-205) on stock torch CPU modules -- the classes only order calls, so their contract
can be checked against hand-written loops without a GPU."""
import types

import pytest
import torch
import torch.nn.functional as F

from fusion_gcn_amd.session.procedures import (DefaultBatchProcessor, DefaultStep, GradientAccumulationBatchProcessor, GraphStep,
                                               MixedPrecisionStep, Step, get_batch_processor_from_config)
from fusion_gcn_amd.session.session import Session


def tiny(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def batch(n=8, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 6, generator=g), torch.randint(0, 3, (n,), generator=g), torch.arange(n)


def test_default_step_is_forward_loss_backward():
    model, (x, y, _) = tiny(), batch()
    step = DefaultStep()
    y_pred, loss = step.forward(model, F.cross_entropy, x, y, loss_quotient=4)
    step.backward(loss)
    want = F.cross_entropy(model(x), y) / 4
    assert torch.equal(y_pred, model(x)) and torch.equal(loss, want)
    grads = torch.autograd.grad(want, list(model.parameters()))
    for p, g in zip(model.parameters(), grads):
        assert torch.allclose(p.grad, g, atol=1e-7)
    assert isinstance(step, Step) and step.reset() is None


def test_accumulating_processor_follows_the_reference_quotient():
    """Micro-batches of size s: every micro loss is CE_mean(micro) / s (batch_train.py:95-96), gradients add up."""
    model, (x, y, idx) = tiny(), batch(8)
    seen = []
    GradientAccumulationBatchProcessor(DefaultStep(), 8, 2).process_single_batch(
        model, F.cross_entropy, x, y, idx, lambda loss, pair, m, i: seen.append((float(loss), tuple(pair[0].shape), i.tolist())))
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    total = sum(F.cross_entropy(model(x[lo:lo + 2]), y[lo:lo + 2]) / 2 for lo in range(0, 8, 2))
    total.backward()
    for g, p in zip(got, model.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-7)
    assert [s[2] for s in seen] == [[0, 1], [2, 3], [4, 5], [6, 7]] and all(s[1] == (2, 3) for s in seen)
    with pytest.raises(AssertionError):
        GradientAccumulationBatchProcessor(DefaultStep(), 8, 3)


def test_dictionary_features_are_sliced_per_modality():
    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(6, 3), torch.nn.Linear(4, 3)

        def forward(self, f):
            return self.a(f["x"]) + self.b(f["z"])
    torch.manual_seed(0)
    model = Two()
    x, y, idx = batch(4)
    feats = {"x": x, "z": torch.randn(4, 4)}
    shapes = []
    GradientAccumulationBatchProcessor(DefaultStep(), 4, 2).process_single_batch(
        model, F.cross_entropy, feats, y, idx, lambda loss, pair, m, i: shapes.append(tuple(pair[0].shape)))
    assert shapes == [(2, 3), (2, 3)] and model.b.weight.grad is not None


def test_processor_selection_from_the_session_config():
    ns = lambda **kw: types.SimpleNamespace(**{"batch_size": 16, "grad_accum_step": 16, "mixed_precision": False, **kw})   # noqa: E731
    p = get_batch_processor_from_config(ns(), {})
    assert isinstance(p, DefaultBatchProcessor) and isinstance(p._step_function, DefaultStep)
    p = get_batch_processor_from_config(ns(mixed_precision=True, grad_accum_step=4), {"batch_size": 8, "grad_accum_step": 2})
    assert isinstance(p, GradientAccumulationBatchProcessor) and isinstance(p._step_function, MixedPrecisionStep)
    assert (p._steps, p._gradient_accumulation_batch_size) == (4, 2)
    p = get_batch_processor_from_config(ns(mixed_precision=True), {"hip_graph": True})
    assert isinstance(p._step_function, GraphStep) and p._step_function.math == "bf16"
    assert isinstance(get_batch_processor_from_config(ns(hip_graph=True), {})._step_function, GraphStep)
    container = {}
    get_batch_processor_from_config(ns(mixed_precision=True), {}).get_state_dict_objects(container)
    assert list(container) == ["loss_scale"] and container["loss_scale"].state_dict() == {}       # the reference's checkpoint key


def test_mixed_precision_step_runs_both_passes_in_the_bf16_mode():
    from fusion_gcn_amd import ops
    seen = []

    class Spy(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            seen.append(("fwd", ops.get_math_mode()))
            return t * 1.0

        @staticmethod
        def backward(ctx, g):
            seen.append(("bwd", ops.get_math_mode()))
            return g
    model, (x, y, _) = tiny(), batch()
    step = MixedPrecisionStep()
    before = ops.get_math_mode()
    y_pred, loss = step.forward(lambda t: Spy.apply(model(t)), F.cross_entropy, x, y)
    step.backward(loss)
    assert seen == [("fwd", "bf16"), ("bwd", "bf16")] and ops.get_math_mode() == before


def test_epoch_loops_match_a_hand_written_loop():
    data = [batch(4, seed=s) for s in range(3)]
    model, ref = tiny(3), tiny(3)
    opt, ref_opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9), torch.optim.SGD(ref.parameters(), 0.1, momentum=0.9)

    class Metrics:
        def __init__(self):
            self.train, self.val = [], []

        def update_training(self, loss, pair, m, idx):
            self.train.append(float(loss))

        def update_validation(self, loss, pair, m, idx):
            self.val.append(float(loss))

        def format_training(self):
            return "t"

        def format_all(self):
            return "a"

    class Progress:
        calls = []

        def update_epoch_mode(self, mode, metrics=None):
            self.calls.append((mode, metrics))
    metrics, progress = Metrics(), Progress()
    proc = DefaultBatchProcessor(DefaultStep())
    Session.train_epoch(proc, model, F.cross_entropy, [(x.double(), y.int(), i) for x, y, i in data], opt, progress, metrics)
    want = []
    for x, y, _ in data:
        ref_opt.zero_grad()
        loss = F.cross_entropy(ref(x), y)
        loss.backward()
        ref_opt.step()
        want.append(float(loss))
    assert metrics.train == pytest.approx(want, abs=1e-7)            # (inputs arrive as float64 / int32: cast like the reference)
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-7)
    Session.validate_epoch(proc, model, F.cross_entropy, data, progress, metrics, mode=2)
    assert not model.training and len(metrics.val) == 3 and all(p.grad is not None for p in model.parameters())
    with torch.no_grad():
        assert metrics.val == pytest.approx([float(F.cross_entropy(ref(x), y)) for x, y, _ in data], abs=1e-6)
    assert progress.calls == [(0, "t")] * 3 + [(2, "a")] * 3


def test_graph_step_needs_the_device_and_evaluates_eagerly():
    model, (x, y, _) = tiny(), batch()
    step = GraphStep()
    with pytest.raises(RuntimeError, match="HIP device"):
        step.forward(model, F.cross_entropy, x, y)
    model.eval()
    y_pred, loss = step.forward(model, F.cross_entropy, x, y)          # evaluation: the eager forward, nothing recorded
    assert torch.equal(y_pred, model(x)) and step.replays == 0
    model.train()
    with torch.no_grad():
        assert torch.equal(step.forward(model, F.cross_entropy, x, y)[0], model(x))
    # a loss that did not come out of a replay is differentiated eagerly
    loss = F.cross_entropy(model(x), y)
    step.backward(loss)
    assert model[0].weight.grad is not None
