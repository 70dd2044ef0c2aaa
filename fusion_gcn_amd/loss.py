"""The step's loss on libfgcn.  The reference builds ``torch.nn.CrossEntropyLoss()`` once per session and hands it to every step
(torch_src/session/session.py:53, session/procedures/step.py:38-46); ``CrossEntropyLoss`` here is that object -- same call
signature, mean reduction, rows labelled outside [0, classes) (torch's ignore_index = -100) do not count -- computed by one
fixed-order kernel each way (``fgcn_cross_entropy_fwd`` / ``_bwd``, include/fgcn.h): with ``data_bn`` on libfgcn as well, every
gradient of a training step is bitwise reproducible.  No fallback: raises without libfgcn / off gfx950.
"""
from __future__ import annotations

import torch

from .block import cross_entropy


class CrossEntropyLoss(torch.nn.Module):
    def forward(self, y_pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        return cross_entropy(y_pred, label)


__all__ = ["CrossEntropyLoss", "cross_entropy"]
