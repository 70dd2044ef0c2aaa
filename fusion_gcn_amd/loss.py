"""The step's loss on libfgcn.  The reference builds ``torch.nn.CrossEntropyLoss()`` once per session and hands it to every step
(torch_src/session/session.py:53, session/procedures/step.py:38-46); ``CrossEntropyLoss`` here is that object -- same call
signature, mean reduction, rows labelled -100 (torch's ignore_index) do not count, any other label outside [0, classes) turns the loss
and the gradients into NaN (torch raises a device assert there) -- computed by one fixed-order kernel each way
(``fgcn_cross_entropy_fwd`` / ``_bwd``, include/fgcn.h).  A step that uses THIS loss (bench.py, GraphStep / the session when the
config's loss is built from this module) has bitwise reproducible gradients throughout, ``data_bn`` being on libfgcn as well; a
session that is handed ``torch.nn.CrossEntropyLoss`` instead runs torch's kernels for the loss.  No fallback: raises without
libfgcn / off gfx950.
"""
from __future__ import annotations

import torch

from .block import cross_entropy


class CrossEntropyLoss(torch.nn.Module):
    def forward(self, y_pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        return cross_entropy(y_pred, label)


__all__ = ["CrossEntropyLoss", "cross_entropy"]
