"""``GCN`` -- ST-GCN / AGCN without the temporal component, for graphs whose nodes are IMU samples (reference
torch_src/models/mmargcn/gcn.py:18-83): BatchNorm1d over the flattened (feature, node) input, ``num_layers`` graph
convolutions (width doubling every third layer), mean over the nodes, ``fc``.  Same constructor and state-dict keys
(``gc<i>.*``, ``bn.*``, ``fc.*``); the layers run node-major on libfgcn kernels (graph_convolution.py), pooling and ``fc`` on
``fgcn_group_mean`` / ``fgcn_rows_gemm``."""
import math
from typing import List, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...block import GroupMeanFunction, LinearFunction, data_bn
from .graph_convolution import AGCNGraphConvolution, STGCNGraphConvolution

_GRAPH_CONVS = {"stgcn": STGCNGraphConvolution, "agcn": AGCNGraphConvolution}


def layer_plan(feature_dim: int, width: int, num_layers: int, extra_top: bool) -> List[Tuple[int, int, bool]]:
    """(in, out, is_input_layer) per graph convolution: the input layer (built without residual and dropout), an optional
    second layer of the same width, then layers whose width doubles at every third position (gcn.py:42-58)."""
    plan = [(feature_dim, width, True)]
    if extra_top:
        plan.append((width, width, False))
    for pos in range(1, num_layers - len(plan) + 1):
        nxt = width * 2 if pos % 3 == 0 else width
        plan.append((width, nxt, False))
        width = nxt
    return plan


class GCN(nn.Module):
    def __init__(self, adj: Union[torch.Tensor, torch.sparse.Tensor], data_shape: tuple, num_classes: int,
                 dropout: float = 0., sparse: bool = False, gc_model: str = "stgcn", num_layers: int = 10,
                 inner_feature_dim: int = 64, include_additional_top_layer: bool = False, without_fc: bool = False):
        super().__init__()
        if num_layers < 2:
            raise AssertionError("num_layers >= 2")
        if gc_model not in _GRAPH_CONVS:
            raise ValueError(f"Model {gc_model} not supported.")
        conv = _GRAPH_CONVS[gc_model]
        feature_dim, num_nodes = data_shape
        self.bn = nn.BatchNorm1d(feature_dim * num_nodes)
        self.layers = []
        for idx, (cin, cout, first) in enumerate(layer_plan(feature_dim, inner_feature_dim, num_layers,
                                                             include_additional_top_layer), start=1):
            layer = conv(cin, cout, adj, sparse=sparse, residual=False) if first else conv(cin, cout, adj, sparse=sparse,
                                                                                          dropout=dropout)
            self.layers.append(layer)
            setattr(self, f"gc{idx}", layer)          # the reference's attribute / state-dict names
        self.fc = None
        if not without_fc:
            self.fc = nn.Linear(self.layers[-1].out_features, num_classes)
            nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_classes))

    def forward(self, x):
        batch_size, feature_dim, num_nodes = x.size()
        # input BatchNorm1d over the (feature, node) channels, statistics across the batch (gcn.py:59-61 of the reference): libfgcn's
        # data_bn kernels with (N, M, T, V, C) = (batch, 1, 1, features, nodes) -- channel index f * nodes + node, as torch.flatten gives
        # it; running statistics and the batch counter are updated like the module's own forward would
        h = data_bn(x.reshape(batch_size, 1, 1, feature_dim, num_nodes), self.bn)       # (B, 1, F, nodes padded to 4)
        h = h[:, 0, :, :num_nodes].permute(0, 2, 1)      # node-major (B, V, F), channels padded to 4
        pad = (-feature_dim) % 4
        h = (F.pad(h, (0, pad)) if pad else h).contiguous()
        for layer in self.layers:
            h = layer(h)
        h = GroupMeanFunction.apply(h)                   # x.mean(-1): mean over the nodes
        if self.fc is not None:
            h = LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)
        return h
