"""Batch data parallelism: one process per GPU, one flat gradient buffer, ONE all-reduce per step.

The reference has no distributed code; what a replica computes is the reference's single-GPU
step (session/procedures/step.py:38-46) on its shard of the clip batch, with per-replica BatchNorm statistics
(like DDP without SyncBN).  After backward the 274 parameter gradients are gathered into one contiguous fp32
buffer (13.9 MB for the 60-class model) with a single multi-tensor copy, exchanged with a single RCCL all-reduce
over xGMI, scaled by 1/world, and handed back to the parameters as views of that buffer (no copy back) — no
per-parameter collectives, no bucketing logic, nothing to overlap it with that would matter (>= 6 ms of compute
per step vs ~0.1 ms of collective).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatGradients:
    """Flat fp32 gradient buffer for the data-parallel exchange."""

    def __init__(self, params: Iterable[torch.nn.Parameter], allow_unused: bool = False):
        self.allow_unused = allow_unused
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        offsets, total = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share one device and dtype")
            offsets.append(total)
            total += (p.numel() + 3) // 4 * 4          # keeps every view 16-byte aligned
        # allow_unused: one "used" slot per parameter + one "anything unused?" slot ride behind the gradients in the SAME allocation,
        # so that all_reduce_mean(restore_unused=True) exchanges gradients and mask in one collective (self.flat stays the gradients)
        self._n_mask = (len(self.params) + 1 + 3) // 4 * 4 if allow_unused else 0
        self._buf = torch.zeros(total + self._n_mask, device=dev, dtype=dt)
        self.flat = self._buf[:total]
        self.unused: List[bool] = [False] * len(self.params)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, offsets)]

    def zero(self) -> None:
        """optimizer.zero_grad(set_to_none=True): backward then writes fresh gradient tensors (no add kernels)."""
        for p in self.params:
            p.grad = None

    def zero_in_place(self) -> None:
        """optimizer.zero_grad(set_to_none=False): every p.grad becomes its (zeroed) view of the flat buffer, whether or not a
        gradient existed (before the first backward, or after a set_to_none zero, there is none -- that is not an error here)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def gather(self) -> None:
        """Copy this step's gradients into the flat buffer (one multi-tensor copy) and re-point p.grad at it."""
        src, dst = [], []
        self.unused = [p.grad is None for p in self.params]     # (allow_unused: all_reduce_mean(restore_unused=True) reads it)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                # torch.optim skips parameters without a gradient; the flat update cannot, so an unused parameter is an
                # error unless the caller opted into "zero gradient" semantics (weight decay / momentum still apply)
                if not self.allow_unused:
                    raise RuntimeError("FlatGradients.gather: a trainable parameter received no gradient this step "
                                       "(unused in the forward?); pass allow_unused=True to treat it as a zero gradient")
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None, always: bool = False, restore_unused: bool = False) -> None:
        """Average gradients across replicas: one collective on the flat buffer.  ``always``: issue the collective for a
        single-rank group too (what a 1-GPU box can prove about the RCCL path: tests/test_rccl_gpu.py).  ``restore_unused`` (with
        ``allow_unused``): a parameter that received no gradient on ANY rank gets ``p.grad = None`` back after the exchange, so a
        torch.optim optimizer skips it exactly as in the single-rank run (no weight decay / momentum on it); one that some rank
        did use keeps the averaged gradient on every rank.  The used-mask travels behind the gradients in the same collective (every
        step, by design: the ranks cannot know beforehand whether another rank had an unused parameter) and is read back with ONE
        device-to-host copy."""
        self.gather()
        if not (dist.is_available() and dist.is_initialized()):
            self._restore_unused(restore_unused, self.unused)
            return
        world = dist.get_world_size(group)
        if world == 1 and not always:
            self._restore_unused(restore_unused, self.unused)
            return
        with_mask = restore_unused and self.allow_unused
        if with_mask:
            n = len(self.params)
            host = torch.tensor([0.0 if u else 1.0 for u in self.unused] + [float(any(self.unused))], dtype=self.flat.dtype)
            self._buf[self.flat.numel():self.flat.numel() + n + 1].copy_(host, non_blocking=True)
        dist.all_reduce(self._buf if with_mask else self.flat, op=dist.ReduceOp.SUM, group=group)
        if world > 1:
            self.flat.mul_(1.0 / world)
        if with_mask:
            mask = self._buf[self.flat.numel():self.flat.numel() + n + 1].tolist()      # the one host sync of this path
            if mask[-1] > 0:
                self._restore_unused(True, [m == 0.0 for m in mask[:-1]])

    def _restore_unused(self, enabled: bool, unused) -> None:
        if enabled and self.allow_unused:
            for p, u in zip(self.params, unused):
                if u:
                    p.grad = None


def shard_batch(n_total: int, rank: int, world: int) -> slice:
    """Contiguous clip shard of rank ``rank``: clips [r*N/P, (r+1)*N/P)  (SURVEY.md §8e)."""
    if n_total % world:
        raise ValueError(f"global batch {n_total} is not divisible by world size {world}")
    per = n_total // world
    return slice(rank * per, (rank + 1) * per)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    # c10d collectives write through the storage and do NOT bump the tensors' version counters (checked on torch 2.10: _version
    # is the same before and after), so everything keyed on parameter versions -- the blocks' packed weights, fops.ParamForms --
    # is told explicitly: versions are incremented (metadata only) and modules with packed forms are marked stale
    with torch.no_grad():
        tensors = list(module.parameters()) + list(module.buffers())
        for t in tensors:
            dist.broadcast(t.detach(), src=src, group=group)
        for t in tensors:
            torch.autograd.graph.increment_version(t)
    for m in module.modules():
        if hasattr(m, "mark_packed_stale"):
            m.mark_packed_stale()
