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
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, offsets)]

    def zero(self) -> None:
        """optimizer.zero_grad(set_to_none=True): backward then writes fresh gradient tensors (no add kernels)."""
        for p in self.params:
            p.grad = None

    def gather(self) -> None:
        """Copy this step's gradients into the flat buffer (one multi-tensor copy) and re-point p.grad at it."""
        src, dst = [], []
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

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Average gradients across replicas: one collective on the flat buffer."""
        self.gather()
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / world)


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
    # in place on t.detach() under no_grad: that alias shares the tensor's version counter, so everything keyed on parameter
    # versions (the blocks' cache of packed weights) sees the change; a write through ``t.data`` would not bump it
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src=src, group=group)
    for m in module.modules():
        if hasattr(m, "_wcache"):
            m._wcache = None
