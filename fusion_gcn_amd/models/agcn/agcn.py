"""``model: agcn`` — the same network under the reference's original 2s-AGCN spelling
(torch_src/models/agcn/agcn.py: unit_tcn :38-52, unit_gcn :55-113, TCN_GCN_unit :116-133, Model :136-191).

State-dict names follow that file: layers ``l1..l10``, learned adjacency ``gcn1.PA``, and — as in the reference —
the constant adjacency ``A`` is NOT part of the state dict (it is rebuilt from the graph).  The reference's
``unit_gcn.forward`` is CUDA-only (``self.A.cuda(x.get_device())``, :97); here the block runs in libfgcn.
``Model(data_shape={"skeleton": (M, T, V, C)}, num_classes, graph, **kwargs)`` with the optional
``adjacency_matrix`` keyword (:144-146).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from ..mmargcn import agcn as _base
from ..mmargcn.agcn import bn_init, conv_branch_init, conv_init  # noqa: F401  (same helpers, same names)
from ...util.partition_strategy import GraphPartitionStrategy


class unit_tcn(_base.TemporalConv):
    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride)
        self.relu = nn.ReLU()          # defined but never applied by the reference either (:46,51-52)


class unit_gcn(_base.GraphConvParams):
    """The adaptive graph convolution's parameters under the 2s-AGCN names: learned adjacency ``PA``; the constant adjacency ``A``
    is a plain attribute, NOT part of the state dict (reference :60-62), so it follows the module through .to() / .cuda() by hand."""

    ADJ_PARAM = "PA"

    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3):
        if (coff_embedding, num_subset) != (4, 3):
            raise ValueError("the HIP AGCN block implements coff_embedding=4, num_subset=3")
        super().__init__(in_channels, out_channels, torch.from_numpy(np.asarray(A, dtype=np.float32)), num_subset)
        self.num_subset, self.inter_c = num_subset, self.embedding_channels
        self.soft, self.relu = nn.Softmax(-2), nn.ReLU()

    def setattr_adjacency(self, adjacency: torch.Tensor) -> None:
        self.PA = nn.Parameter(torch.full_like(adjacency, 1e-6))
        self.A = adjacency

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.A = fn(self.A)
        return out


class TCN_GCN_unit(_base.SpatialTemporalConv):
    _ADJ_PARAM = "gcn1.PA"
    _GCN, _TCN = unit_gcn, unit_tcn

    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, **kw):
        super().__init__(in_channels, out_channels, A, stride=stride, residual=residual, **kw)
        self.relu = nn.ReLU()

    def _adj_a(self) -> torch.Tensor:
        return self.gcn1.A


# (in, out, stride) of l1..l10; l1 has no block residual
_LAYERS = ((None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1), (128, 256, 2),
           (256, 256, 1), (256, 256, 1))


class Model(nn.Module):
    def __init__(self, data_shape, num_classes, graph, **kwargs):
        super().__init__()
        num_persons, _, num_joints, num_channels = data_shape["skeleton"]      # (persons, frames, joints, channels)
        adj = kwargs.get("adjacency_matrix")
        if adj is None:
            adj = GraphPartitionStrategy().get_adjacency_matrix_array(graph)
        kw = {k: kwargs.get(k, default) for k, default in (("static_adjacency", False), ("fused_spatial", True))}
        self.data_bn = nn.BatchNorm1d(num_persons * num_channels * num_joints)
        bn_init(self.data_bn, 1)
        for i, (cin, cout, stride) in enumerate(_LAYERS, start=1):
            setattr(self, f"l{i}", TCN_GCN_unit(num_channels if cin is None else cin, cout, adj, stride=stride,
                                                residual=cin is not None, **kw))
        self.fc = nn.Linear(_LAYERS[-1][1], num_classes)
        nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_classes))

    _blocks_input = _base.Model._blocks_input
    _bump_batch_counters = _base.Model._bump_batch_counters
    prepare_recording = _base.Model.prepare_recording
    recording_pins = _base.Model.recording_pins

    def forward(self, x):
        clips = x.size(0)
        h = self._blocks_input(x)
        self._bump_batch_counters()
        _base.refresh_packed_weights(self)
        for i in range(1, len(_LAYERS) + 1):
            h = getattr(self, f"l{i}")(h)
        h = _base.GroupMeanFunction.apply(h.view(clips, -1, h.size(-1)))
        return _base.LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)   # fc on the row GEMM
