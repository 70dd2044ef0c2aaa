"""The step after the hot path: one fused parameter update over flat buffers.

The reference builds ``torch.optim.{SGD, Adam, AdamW}(model.parameters(), lr, **optimizer_args)`` and an optional
``torch.optim.lr_scheduler`` over it (torch_src/session_helper.py:48-89; ADAM + weight_decay 0.01 + ``cawr`` in
config/utd-mhad/skeleton/agcn.yaml:15-22) and calls ``optimizer.step()`` after every batch (session/session.py:176-183):
one small launch chain per parameter tensor, 274 tensors.  ``FlatOptimizer`` keeps that interface -- it *is* a
``torch.optim.Optimizer`` (``param_groups[0]["lr"]``, ``zero_grad``, ``state_dict``; torch's LR schedulers drive it
unchanged) -- but re-homes every trainable parameter into ONE contiguous float32 buffer, shares the flat gradient buffer of
the data-parallel exchange (``dp.FlatGradients``), and applies the update with a single libfgcn launch
(``fgcn_optim_step``, include/fgcn.h) whose arithmetic follows torch's formulas operation by operation.

No fallback: without libfgcn.so / off gfx950 ``step()`` raises ``FgcnError``.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from . import _lib
from .dp import FlatGradients

KINDS = {"SGD": 0, "ADAM": 1, "ADAMW": 2}     # FGCN_OPT_* (include/fgcn.h); names as in session_helper.available_optimizers


class FlatOptimizer(torch.optim.Optimizer):
    """``FlatOptimizer(model.parameters(), "ADAM", lr, weight_decay=0.01)`` == ``create_optimizer("ADAM", model, lr, ...)``.

    Supported ``optimizer_args`` (torch names and defaults): SGD ``momentum, dampening, weight_decay, nesterov``;
    ADAM / ADAMW ``betas, eps, weight_decay`` (AdamW's default decay is 0.01).  ``amsgrad`` / ``maximize`` and more than
    one parameter group are not built (the reference uses neither) and raise.
    ``grads``: an existing ``FlatGradients`` over the same parameters (the data-parallel buffer) to share.
    ``allow_unused``: a trainable parameter without a gradient in a step counts as a zero gradient instead of an error
    (forwarded to the FlatGradients this optimizer creates).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], name: str = "ADAM", lr: float = 1e-3, *,
                 grads: Optional[FlatGradients] = None, allow_unused: bool = False, **optimizer_args):
        kind = name.upper()
        if kind not in KINDS:
            raise ValueError("Unsupported optimizer: " + kind + " (SGD | ADAM | ADAMW)")
        if optimizer_args.get("amsgrad") or optimizer_args.get("maximize"):
            raise NotImplementedError("amsgrad / maximize are not built")
        defaults = dict(lr=lr, weight_decay=0.01 if kind == "ADAMW" else 0.0)
        if kind == "SGD":
            defaults.update(momentum=0.0, dampening=0.0, nesterov=False)
        else:
            defaults.update(betas=(0.9, 0.999), eps=1e-8)
        unknown = set(optimizer_args) - set(defaults) - {"amsgrad", "maximize"}
        if unknown:
            raise TypeError(f"{kind}: unexpected optimizer_args {sorted(unknown)}")
        defaults.update({k: v for k, v in optimizer_args.items() if k in defaults})
        if lr < 0 or defaults["weight_decay"] < 0:
            raise ValueError("negative lr / weight_decay")
        if kind == "SGD" and defaults["nesterov"] and (defaults["momentum"] <= 0 or defaults["dampening"] != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        params = list(params)
        if params and isinstance(params[0], dict):
            raise NotImplementedError("FlatOptimizer takes one parameter group (as the reference's create_optimizer)")
        super().__init__(params, defaults)
        self.kind = kind
        self.grads = grads if grads is not None else FlatGradients(self.param_groups[0]["params"], allow_unused=allow_unused)
        self.params = self.grads.params                      # trainable parameters, in order
        if grads is not None and [id(p) for p in self.params] != [id(p) for p in self.param_groups[0]["params"] if p.requires_grad]:
            raise ValueError("the shared FlatGradients must cover the same parameters in the same order")
        # one contiguous home for the parameter values, laid out like the gradient buffer (16-byte aligned views)
        self.flat = torch.zeros_like(self.grads.flat)
        with torch.no_grad():
            for p, gv in zip(self.params, self.grads.views):
                off = gv.storage_offset()
                home = self.flat[off:off + p.numel()].view_as(p)
                home.copy_(p)
                p.data = home
        need1 = kind != "SGD" or defaults["momentum"] != 0
        self.state1 = torch.zeros_like(self.flat) if need1 else None       # momentum buffer / exp_avg
        self.state2 = torch.zeros_like(self.flat) if kind != "SGD" else None   # exp_avg_sq
        self.steps = 0
        self.grad_scale = 1.0     # set to 1/world when the flat gradients hold an un-averaged all-reduce sum

    def zero_grad(self, set_to_none: bool = True) -> None:
        if set_to_none:
            self.grads.zero()
        else:
            self.grads.zero_in_place()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_homes()
        self.grads.gather()                   # p.grad -> the flat buffer (no copy when they already are views of it)
        g = self.param_groups[0]
        lib = _lib.load()
        if not self.flat.is_cuda:
            raise _lib.FgcnError("FlatOptimizer.step needs the parameters on an MI355X (no CPU fallback)")
        self.steps += 1
        b1, b2 = g.get("betas", (0.0, 0.0))
        s1 = self.state1.data_ptr() if self.state1 is not None else None
        s2 = self.state2.data_ptr() if self.state2 is not None else None
        rc = lib.fgcn_optim_step(self.flat.data_ptr(), self.grads.flat.data_ptr(), s1, s2, self.flat.numel(),
                                 KINDS[self.kind], float(g["lr"]), float(g["weight_decay"]), float(self.grad_scale),
                                 float(b1), float(b2), float(g.get("eps", 0.0)), float(g.get("momentum", 0.0)),
                                 float(g.get("dampening", 0.0)), int(bool(g.get("nesterov", False))), self.steps,
                                 torch.cuda.current_stream(self.flat.device).cuda_stream)
        _lib.check(rc, "fgcn_optim_step")
        # the kernel wrote through raw pointers: tell autograd (and everything keyed on tensor versions, like the blocks'
        # cache of packed weights) that every parameter changed in place -- metadata only, no launches
        for p in self.params:
            torch.autograd.graph.increment_version(p)
        return loss

    def _check_homes(self) -> None:
        """Every parameter must still live in the flat buffer: a later model.to() / .float() / load_state_dict(assign=True) /
        ``p.data = ...`` gives it new storage, and the fused update would then train an orphaned copy while the model's weights
        stay frozen.  Pointer comparisons only (no launches).  Create the optimizer after the model's final .to() / cast."""
        base = self.flat.data_ptr()
        for p, v in zip(self.params, self.grads.views):
            if p.data_ptr() != base + 4 * v.storage_offset():
                raise _lib.FgcnError("FlatOptimizer: a parameter no longer aliases the flat parameter buffer (the model was moved, "
                                     "cast or re-assigned after the optimizer was created); build the optimizer after the final "
                                     ".to() / cast")

    # ---- torch.optim state-dict layout (per-parameter entries are views of the flat state) ---------------------------------
    def _views(self, flat: torch.Tensor):
        return [flat[v.storage_offset():v.storage_offset() + p.numel()].view_as(p) for p, v in zip(self.params, self.grads.views)]

    def _slots(self):
        """Position of every trainable parameter in ``param_groups[0]["params"]`` (frozen parameters keep their slot, stateless)."""
        where = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
        return [where[id(p)] for p in self.params]

    def state_dict(self) -> Dict:
        """Same layout as the torch optimizer of that name: {"state": {i: {...}}, "param_groups": [...]}."""
        state = {}
        if self.steps:
            s1 = self._views(self.state1) if self.state1 is not None else None
            s2 = self._views(self.state2) if self.state2 is not None else None
            for i, slot in enumerate(self._slots()):      # torch's layout: state index = position in param_groups[0]["params"]
                if self.kind == "SGD":
                    state[slot] = {"momentum_buffer": s1[i].clone() if s1 is not None else None}
                else:
                    state[slot] = {"step": torch.tensor(float(self.steps)), "exp_avg": s1[i].clone(), "exp_avg_sq": s2[i].clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.param_groups[0]["params"])))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        group = sd["param_groups"][0]
        for k, v in group.items():
            if k != "params":
                self.param_groups[0][k] = v
        st = sd.get("state", {})
        self.steps = 0
        if st:
            s1 = self._views(self.state1) if self.state1 is not None else None
            s2 = self._views(self.state2) if self.state2 is not None else None
            with torch.no_grad():
                for i, slot in enumerate(self._slots()):
                    e = st[slot] if slot in st else st[str(slot)]
                    if self.kind == "SGD":
                        if s1 is not None and e.get("momentum_buffer") is not None:
                            s1[i].copy_(e["momentum_buffer"])
                            self.steps = max(self.steps, 1)
                    else:
                        s1[i].copy_(e["exp_avg"])
                        s2[i].copy_(e["exp_avg_sq"])
                        self.steps = int(e["step"])


def create_optimizer(name: str, model: torch.nn.Module, lr: float, **optimizer_args) -> FlatOptimizer:
    """Signature of the reference's session_helper.create_optimizer (torch_src/session_helper.py:80-84)."""
    return FlatOptimizer(model.parameters(), name, lr, **optimizer_args)
