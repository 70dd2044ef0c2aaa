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


class unit_gcn(_base._KernelBacked):
    """Parameter container of the adaptive graph convolution under the 2s-AGCN names (``PA``, ``conv_a/b/d.<k>``, ``down``, ``bn``)."""

    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3):
        super().__init__()
        if (coff_embedding, num_subset) != (4, 3):
            raise ValueError("the HIP AGCN block implements coff_embedding=4, num_subset=3")
        adjacency = torch.from_numpy(np.asarray(A, dtype=np.float32))
        self.num_subset, self.inter_c = num_subset, out_channels // coff_embedding
        self.A = adjacency                                   # plain attribute: not in the state dict (reference :60-62)
        self.PA = nn.Parameter(torch.full_like(adjacency, 1e-6))
        self.adj_c = [None] * num_subset
        widths = {"conv_a": self.inter_c, "conv_b": self.inter_c, "conv_d": out_channels}
        for name, width in widths.items():
            setattr(self, name, nn.ModuleList(nn.Conv2d(in_channels, width, 1) for _ in range(num_subset)))
        self.down = (nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
                     if in_channels != out_channels else (lambda x: x))
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft, self.relu = nn.Softmax(-2), nn.ReLU()
        # the reference's initialisation (:84-93): kaiming fan-out convs, unit BatchNorms, the output BatchNorm and PA at
        # 1e-6, conv_d with the branch-count variance
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for conv in self.conv_d:
            conv_branch_init(conv, num_subset)

    def _apply(self, fn, *args, **kwargs):
        # keep the non-registered constant adjacency on the module's device (.cuda() / .to())
        out = super()._apply(fn, *args, **kwargs)
        self.A = fn(self.A)
        return out


class TCN_GCN_unit(_base.SpatialTemporalConv):
    _ADJ_PARAM = "gcn1.PA"

    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, **kw):
        nn.Module.__init__(self)
        self.gcn1 = unit_gcn(in_channels, out_channels, A)
        self.tcn1 = unit_tcn(out_channels, out_channels, stride=stride)
        self.relu = nn.ReLU()
        self.out_channels = out_channels
        if not residual:
            self.residual = lambda x: 0
            res = "none"
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = lambda x: x
            res = "identity"
        else:
            self.residual = unit_tcn(in_channels, out_channels, kernel_size=1, stride=stride)
            res = "conv"
        self.cfg = _base.BlockConfig(cin=in_channels, cout=out_channels, stride=stride, residual=res,
                                     has_down=in_channels != out_channels, **kw)
        self.cfg.validate()
        self._wcache = None

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

    def forward(self, x):
        clips = x.size(0)
        h = self._blocks_input(x)
        self._bump_batch_counters()
        for i in range(1, len(_LAYERS) + 1):
            h = getattr(self, f"l{i}")(h)
        h = _base.GroupMeanFunction.apply(h.view(clips, -1, h.size(-1)))
        return _base.LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)   # fc on the row GEMM
