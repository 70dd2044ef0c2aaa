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
    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3):
        super().__init__()
        if coff_embedding != 4 or num_subset != 3:
            raise ValueError("the HIP AGCN block implements coff_embedding=4, num_subset=3")
        inter_channels = out_channels // coff_embedding
        self.inter_c = inter_channels
        self.PA = nn.Parameter(torch.from_numpy(A.astype(np.float32)))
        nn.init.constant_(self.PA, 1e-6)
        self.A = torch.from_numpy(A.astype(np.float32))     # plain attribute: not in the state dict
        self.num_subset = num_subset
        self.adj_c = [None] * num_subset
        self.conv_a = nn.ModuleList()
        self.conv_b = nn.ModuleList()
        self.conv_d = nn.ModuleList()
        for _ in range(self.num_subset):
            self.conv_a.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_b.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_d.append(nn.Conv2d(in_channels, out_channels, 1))
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = lambda x: x
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)
        self.relu = nn.ReLU()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for i in range(self.num_subset):
            conv_branch_init(self.conv_d[i], self.num_subset)

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


class Model(nn.Module):
    def __init__(self, data_shape, num_classes, graph, **kwargs):
        super().__init__()
        # data_shape = (num_persons, num_frames, num_joints, num_channels)
        num_persons, _, num_joints, num_channels = data_shape["skeleton"]
        adj = kwargs.get("adjacency_matrix", None)
        if adj is None:
            adj = GraphPartitionStrategy().get_adjacency_matrix_array(graph)
        kw = dict(static_adjacency=kwargs.get("static_adjacency", False), fused_spatial=kwargs.get("fused_spatial", True))
        self.data_bn = nn.BatchNorm1d(num_persons * num_channels * num_joints)
        self.l1 = TCN_GCN_unit(num_channels, 64, adj, residual=False, **kw)
        self.l2 = TCN_GCN_unit(64, 64, adj, **kw)
        self.l3 = TCN_GCN_unit(64, 64, adj, **kw)
        self.l4 = TCN_GCN_unit(64, 64, adj, **kw)
        self.l5 = TCN_GCN_unit(64, 128, adj, stride=2, **kw)
        self.l6 = TCN_GCN_unit(128, 128, adj, **kw)
        self.l7 = TCN_GCN_unit(128, 128, adj, **kw)
        self.l8 = TCN_GCN_unit(128, 256, adj, stride=2, **kw)
        self.l9 = TCN_GCN_unit(256, 256, adj, **kw)
        self.l10 = TCN_GCN_unit(256, 256, adj, **kw)
        self.fc = nn.Linear(256, num_classes)
        nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_classes))
        bn_init(self.data_bn, 1)

    _blocks_input = _base.Model._blocks_input
    _bump_batch_counters = _base.Model._bump_batch_counters

    def forward(self, x):
        N, M, T, V, C = x.size()
        h = self._blocks_input(x)
        self._bump_batch_counters()
        for layer in (self.l1, self.l2, self.l3, self.l4, self.l5, self.l6, self.l7, self.l8, self.l9, self.l10):
            h = layer(h)
        c_new = h.size(-1)
        h = _base.GroupMeanFunction.apply(h.view(N, -1, c_new))
        return _base.LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)   # fc on the row GEMM
