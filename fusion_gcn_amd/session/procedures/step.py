"""One training / evaluation step of a model: the reference's ``Step`` objects (torch_src/session/procedures/step.py:7-78) and the
MI355X-native one, ``GraphStep``.

Interface (same names, argument meaning and call order as the reference; its batch processors and epoch loops drive these
unchanged):  ``forward(model, loss_function, features, label, loss_quotient=1) -> (y_pred, loss)``, ``backward(loss)``,
``run_optimizer_step(optimizer)``, ``reset()``, ``get_state_dict_objects(container)``.

  DefaultStep          step.py:38-52: eager launches in the current math mode (``ops.set_math_mode``; bf16x3 = f32-accurate).
  MixedPrecisionStep   step.py:55-78 (``autocast`` + ``GradScaler``) -> the bf16 math mode (BASELINE config 5): bf16 MFMA operands,
                       float32 accumulation, storage and gradients -- nothing leaves the float32 range, so there is no loss scale.
  GraphStep            forward + loss + backward of a batch shape recorded ONCE into a HIP graph and replayed: a step is ~330
                       kernel launches (MS-G3D: ~1600), whose host cost exceeds the GPU time below ~16 clips per GPU
                       (DESIGN.md sections 5, 8: 8-clip AGCN shard 2x, MS-G3D 2x).

No fallback: the models these run raise without libfgcn / off gfx950, and ``GraphStep`` raises without a HIP device.
"""
from __future__ import annotations

import abc
from typing import Dict, Optional, Union

import torch
import torch.distributed as dist

from ... import ops
from ...dp import FlatGradients

Features = Union[torch.Tensor, Dict[str, torch.Tensor]]


class Step(abc.ABC):
    """Forward and backward pass of one (micro-)batch."""

    @abc.abstractmethod
    def forward(self, model: torch.nn.Module, loss_function, features: Features, label: torch.Tensor, loss_quotient: int = 1):
        """-> (y_pred, loss) with ``loss = loss_function(y_pred, label) / loss_quotient``."""

    @abc.abstractmethod
    def backward(self, loss: torch.Tensor) -> None:
        ...

    _dp_grads: Optional[FlatGradients] = None
    _dp_key: Optional[tuple] = None

    def run_optimizer_step(self, optimizer):
        """``optimizer.step()`` -- after ONE all-reduce (mean) of the flat gradient buffer when a process group with more than one
        rank is active: ``data.ClipBatches`` shards every batch per rank, so without the exchange the replicas would drift apart
        silently.  The buffer is the optimizer's own (``optim.FlatOptimizer.grads``) or one built lazily over its parameters."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            grads = getattr(optimizer, "grads", None)
            if not isinstance(grads, FlatGradients):
                params = [p for g in optimizer.param_groups for p in g["params"]]
                key = (id(optimizer),) + tuple((id(p), p.device) for p in params)
                if self._dp_grads is None or self._dp_key != key:   # another optimizer / model on this Step object, or moved parameters
                    # (torch.optim skips parameters without a gradient: they travel as zeros, and one that NO rank used gets
                    # p.grad = None back after the exchange -- the optimizer then skips it as in the single-rank run)
                    self._dp_grads, self._dp_key = FlatGradients(params, allow_unused=True), key
                grads = self._dp_grads
                grads.all_reduce_mean(restore_unused=True)
            else:
                grads.all_reduce_mean()
        return optimizer.step()

    def reset(self) -> None:
        self._dp_grads, self._dp_key = None, None

    def get_state_dict_objects(self, object_container: dict) -> None:
        """Objects whose state belongs into a checkpoint are added to ``object_container`` (name -> object with state_dict())."""


class DefaultStep(Step):
    def forward(self, model, loss_function, features, label, loss_quotient: int = 1):
        y_pred = model(features)
        return y_pred, loss_function(y_pred, label) / loss_quotient

    def backward(self, loss):
        loss.backward()


class _NoLossScale:
    """Checkpoint stand-in for the reference's GradScaler entry (``loss_scale``): float32 gradients need no scale; loading a
    reference checkpoint's scaler state is accepted and ignored."""

    def state_dict(self):
        return {}

    def load_state_dict(self, _state):
        pass


class MixedPrecisionStep(Step):
    """The bf16 math mode around forward AND backward (the mode is read when a kernel is launched)."""
    MODE = "bf16"

    def __init__(self):
        self._inner = DefaultStep()
        self._loss_scale = _NoLossScale()

    def forward(self, model, loss_function, features, label, loss_quotient: int = 1):
        with ops.math_mode(self.MODE):
            return self._inner.forward(model, loss_function, features, label, loss_quotient)

    def backward(self, loss):
        with ops.math_mode(self.MODE):
            self._inner.backward(loss)

    def reset(self):
        super().reset()
        self._inner = DefaultStep()

    def get_state_dict_objects(self, object_container: dict):
        object_container["loss_scale"] = self._loss_scale


def _signature(features: Features, label: torch.Tensor):
    one = lambda t: (tuple(t.shape), t.dtype)      # noqa: E731
    f = tuple((k, *one(v)) for k, v in sorted(features.items())) if isinstance(features, dict) else one(features)
    return f, one(label)


class _Recorded:
    """One recorded batch shape: the graph, its static inputs / outputs and the fresh gradient tensors it writes."""
    __slots__ = ("graph", "features", "label", "y_pred", "loss", "fresh", "views", "homes", "pins")


class GraphStep(Step):
    """``GraphStep()`` in place of ``DefaultStep()``: the first training batch of every (model, batch shape, loss_quotient, math
    mode) is recorded -- forward, loss, backward and the accumulation of the gradients into ONE flat float32 buffer -- and that
    and every later batch of the shape is a copy of the inputs into the graph's static tensors + one graph launch.

    Semantics kept from the eager step:
      * gradients ACCUMULATE over replays (``GradientAccumulationBatchProcessor``'s micro-steps) until the caller's
        ``optimizer.zero_grad()``: ``p.grad`` is a view of the flat buffer after a step; when ``forward`` finds every ``p.grad`` None
        (zero_grad(set_to_none=True), torch's default) the buffer is cleared first, gradients that are neither None nor these
        views are an error;
      * ``backward(loss)`` of the loss ``forward`` just returned is a no-op (the replay already ran it); any other loss is
        differentiated eagerly;
      * evaluation (``model.eval()`` or under ``torch.no_grad()``) is the eager forward;
      * BatchNorm running statistics / counters advance once per step: the warm-up and the verification replay that precede
        a recording are rolled back.
    ``max_shapes`` recordings are kept (a ragged last batch is a second shape); beyond that the oldest is dropped and recorded again
    when its shape comes back.  ``grads``: a ``dp.FlatGradients`` to share with ``optim.FlatOptimizer`` / the data-parallel all-reduce (created on first use
    otherwise; ``.grads`` afterwards).  ``math``: run in this math mode instead of the current one.  ``verify``: after recording,
    replay once and require the eager step's loss (1e-6) and flat gradient (1e-5 rel-L2; libfgcn's kernels are bitwise reproducible) -- turns a
    recording invalidated by foreign stream use (see below) into an error instead of wrong gradients.

    Stream rule: everything runs on one stream owned by the step, joined to the caller's current stream on both sides.  Autograd
    pins a parameter's gradient accumulation to the stream of the accumulator node's creation, and that node lives as long as ANY
    autograd graph of the model is referenced (a held loss tensor); an accumulator pinned elsewhere invalidates later recordings
    (tools/probes/msg3d_graph_probe.py).  Do not interleave eager backward passes of the same model on other streams.
    """

    def __init__(self, grads: Optional[FlatGradients] = None, math: Optional[str] = None, verify: bool = True,
                 data_parallel: bool = True, max_shapes: int = 4):
        self.grads = grads
        self.math = math
        self.verify = verify
        self.data_parallel = data_parallel
        self.max_shapes = max_shapes          # recordings kept (each owns the activations of a step in its private pool): oldest dropped
        self._eager = DefaultStep()
        self._recorded: Dict[tuple, _Recorded] = {}
        self._stream: Optional[torch.cuda.Stream] = None
        self._done: Optional[torch.Tensor] = None
        self.replays = 0

    # ---- Step interface -------------------------------------------------------------------------------------------------------
    def forward(self, model, loss_function, features, label, loss_quotient: int = 1):
        mode = self.math or ops.get_math_mode()
        if not (model.training and torch.is_grad_enabled()):
            with ops.math_mode(mode):
                return self._eager.forward(model, loss_function, features, label, loss_quotient)
        if not label.is_cuda:
            raise RuntimeError("GraphStep needs the batch on the HIP device (no CPU path)")
        if self.grads is None:
            self.grads = FlatGradients(model.parameters())
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=label.device)
        key = (id(model), id(loss_function), _signature(features, label), loss_quotient, mode)
        caller = torch.cuda.current_stream(label.device)
        self._stream.wait_stream(caller)
        with torch.cuda.stream(self._stream), ops.math_mode(mode):
            self._adopt_gradients()
            rec = self._recorded.get(key)
            if rec is not None and rec.homes != self._homes(model):
                rec = None                  # a parameter / buffer moved (model.to(), a new optimizer's flat home): record again
            if rec is None:
                self._recorded.pop(key, None)
                while len(self._recorded) >= max(1, self.max_shapes):
                    self._recorded.pop(next(iter(self._recorded)))
                rec = self._recorded[key] = self._record(model, loss_function, features, label, loss_quotient)
            self._load(rec, features, label)
            rec.graph.replay()
            self.replays += 1
            y_pred, loss = rec.y_pred.clone(), rec.loss.clone()      # the static outputs are overwritten by the next replay
        caller.wait_stream(self._stream)
        for p, v in zip(self.grads.params, self.grads.views):
            p.grad = v
        self._done = loss
        return y_pred, loss

    def backward(self, loss):
        if loss is self._done:
            self._done = None
            return
        with ops.math_mode(self.math or ops.get_math_mode()):
            loss.backward()

    def run_optimizer_step(self, optimizer):
        """One all-reduce of the flat gradient buffer first when a process group with more than one rank exists (batch data
        parallelism, dp.py; ``data_parallel=False`` leaves the exchange to the caller)."""
        if self.data_parallel and self.grads is not None:
            self.grads.all_reduce_mean()
        return optimizer.step()

    def reset(self):
        super().reset()
        self._recorded.clear()
        self._done = None

    # ---- internals --------------------------------------------------------------------------------------------------------------
    def _adopt_gradients(self) -> None:
        g = self.grads
        held = [p.grad for p in g.params]
        if all(h is None for h in held):
            g.flat.zero_()
        elif not all(h is not None and h.data_ptr() == v.data_ptr() for h, v in zip(held, g.views)):
            raise RuntimeError("GraphStep: parameter gradients that are neither None nor views of the step's flat buffer (an eager "
                               "backward in between?); call optimizer.zero_grad() before the step")

    @staticmethod
    def _homes(model) -> tuple:
        """Addresses the recorded kernels read parameters and buffers from."""
        return tuple(t.data_ptr() for t in model.parameters()) + tuple(t.data_ptr() for t in model.buffers())

    @staticmethod
    def _pins(model) -> list:
        """Objects whose device memory the recorded kernels reach through raw pointers without the graph's pool owning it: the
        packed weight sets and re-pack tables of the modules (``recording_pins`` hook).  A recording holds them for its own
        lifetime: a later forward in another math mode, a broadcast that drops a block's packed set or a rebuilt re-pack plan
        then leaves an older recording's buffers alive (and still re-packed from the current parameters by its own recorded
        launch) instead of freed under it."""
        pins = []
        for m in model.modules():
            hook = getattr(m, "recording_pins", None)
            if callable(hook):
                pins.extend(hook())
        return pins

    @staticmethod
    def _mark_stale(model) -> None:
        """Packed / split weight forms cached per parameter version are rebuilt inside the step (what an optimizer update in
        front of it causes): the re-packing launch becomes part of the recording and every replay packs the CURRENT values."""
        for m in model.modules():
            if hasattr(m, "mark_packed_stale"):
                m.mark_packed_stale()

    @staticmethod
    def _load(rec: _Recorded, features, label) -> None:
        if isinstance(features, dict):
            for k, t in features.items():
                rec.features[k].copy_(t, non_blocking=True)
        else:
            rec.features.copy_(features, non_blocking=True)
        rec.label.copy_(label, non_blocking=True)

    def _eager_step(self, model, loss_function, rec, loss_quotient):
        for p in self.grads.params:
            p.grad = None
        self._mark_stale(model)
        y_pred = model(rec.features)
        loss = loss_function(y_pred, rec.label) / loss_quotient
        loss.backward()
        return y_pred, loss

    def _record(self, model, loss_function, features, label, loss_quotient) -> _Recorded:
        g = self.grads
        rec = _Recorded()
        rec.features = ({k: t.detach().clone() for k, t in features.items()} if isinstance(features, dict) else features.detach().clone())
        rec.label = label.detach().clone()
        rec.homes = self._homes(model)
        state = [b.detach().clone() for b in model.buffers()]          # BatchNorm running statistics / counters
        accumulated = g.flat.clone()
        holders = [p.grad for p in g.params]

        def roll_back():
            with torch.no_grad():
                for b, s in zip(model.buffers(), state):
                    b.copy_(s)

        # warm-up = the reference result: every lazily built buffer (packed weight forms, workspaces) exists afterwards
        _, loss = self._eager_step(model, loss_function, rec, loss_quotient)
        want_loss = loss.detach().clone()
        want = [None if p.grad is None else p.grad.detach().clone() for p in g.params]
        if not g.allow_unused and any(w is None for w in want):
            raise RuntimeError("GraphStep: a trainable parameter received no gradient (unused in the forward?); share a "
                               "FlatGradients(..., allow_unused=True) to treat it as a zero gradient")
        del loss
        roll_back()
        for m in model.modules():                   # what a forward after a parameter update would build lazily and a recording
            if hasattr(m, "prepare_recording"):     # cannot contain (device tables uploaded from the host)
                m.prepare_recording()
        for p in g.params:
            p.grad = None
        rec.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(rec.graph, stream=self._stream):
            rec.y_pred, rec.loss = self._eager_step(model, loss_function, rec, loss_quotient)
            rec.y_pred, rec.loss = rec.y_pred.detach(), rec.loss.detach()
            pairs = [(v, p.grad) for p, v in zip(g.params, g.views) if p.grad is not None]
            rec.views, rec.fresh = [v for v, _ in pairs], [f for _, f in pairs]
            torch._foreach_add_(rec.views, rec.fresh)
        rec.pins = self._pins(model)
        # (a step with active dropout draws other masks on every run -- torch advances the recorded generator offset per replay --
        # so there is nothing to compare it with)
        random_masks = any(isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.p > 0 for m in model.modules())
        if self.verify and not random_masks:
            g.flat.zero_()
            rec.graph.replay()
            # the loss must come back exactly (to float rounding), the flat gradient to 1e-5 in rel-L2: libfgcn's kernels are bitwise
            # reproducible, but torch's own ops in the step need not be (MIOpen's BatchNorm backward moves data_bn.weight by ~1e-3 of
            # its size when another process shares the device) -- an invalid recording is off by orders of magnitude, not by that
            bad = float((rec.loss - want_loss).abs()) > 1e-6 * float(want_loss.abs()) + 1e-30
            num = den = 0.0
            for v, w in zip(g.views, want):
                if w is not None:
                    num += float((v - w).double().pow(2).sum())
                    den += float(w.double().pow(2).sum())
            bad = bad or not (num <= 1e-10 * den + 1e-60)
            roll_back()
            if bad:
                raise RuntimeError("GraphStep: the recorded step does not reproduce the eager one (gradient accumulators pinned to "
                                   "another stream by a held autograd graph? see the class docstring)")
        with torch.no_grad():
            g.flat.copy_(accumulated)
        for p, h in zip(g.params, holders):
            p.grad = h
        return rec
