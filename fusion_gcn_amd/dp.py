"""Batch data parallelism: one process per GPU, one flat gradient buffer, ONE all-reduce per step.

The reference has no distributed code; what a replica computes is the reference's single-GPU
step (session/procedures/step.py:38-46) on its shard of the clip batch, with per-replica BatchNorm statistics
(like DDP without SyncBN).  All 274 parameter gradients live as views into one contiguous fp32 buffer
(13.9 MB for the 60-class model) so the exchange is a single RCCL all-reduce over xGMI followed by a 1/world
scale — no per-parameter collectives, no bucketing logic, nothing to overlap it with that would matter
(>= 6 ms of compute per step vs ~0.1 ms of collective).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatGradients:
    """Owns a flat fp32 gradient buffer; every ``p.grad`` is a view into it (autograd accumulates in place)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # 4-element alignment keeps every view 16-byte aligned
        self.offsets, total = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share one device and dtype")
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.attach()

    def attach(self) -> None:
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self) -> None:
        """Replaces optimizer.zero_grad(): one memset, gradient views stay attached."""
        self.flat.zero_()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Average gradients across replicas: one collective on the flat buffer."""
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
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
