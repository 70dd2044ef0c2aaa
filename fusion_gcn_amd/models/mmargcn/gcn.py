"""``GCN`` -- ST-GCN / AGCN without the temporal component, for graphs whose nodes are IMU samples (reference
torch_src/models/mmargcn/gcn.py:18-83): BatchNorm1d over the flattened (feature, node) input, ``num_layers`` graph
convolutions (width doubling every third layer), mean over the nodes, ``fc``.  Same constructor and state-dict keys
(``gc<i>.*``, ``bn.*``, ``fc.*``); the layers run node-major on libfgcn kernels (graph_convolution.py), pooling and ``fc`` on
``fgcn_group_mean`` / ``fgcn_rows_gemm``."""
import math
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...block import GroupMeanFunction, LinearFunction
from .graph_convolution import AGCNGraphConvolution, STGCNGraphConvolution


class GCN(nn.Module):
    def __init__(self, adj: Union[torch.Tensor, torch.sparse.Tensor], data_shape: tuple, num_classes: int,
                 dropout: float = 0., sparse: bool = False, gc_model: str = "stgcn", num_layers: int = 10,
                 inner_feature_dim: int = 64, include_additional_top_layer: bool = False, without_fc: bool = False):
        super().__init__()
        assert num_layers >= 2
        if gc_model == "stgcn":
            gc = STGCNGraphConvolution
        elif gc_model == "agcn":
            gc = AGCNGraphConvolution
        else:
            raise ValueError(f"Model {gc_model} not supported.")
        feature_dim, num_nodes = data_shape
        self.layers = [gc(feature_dim, inner_feature_dim, adj, sparse=sparse, residual=False)]
        if include_additional_top_layer:
            self.layers.append(gc(inner_feature_dim, inner_feature_dim, adj, sparse=sparse, dropout=dropout))
        k = 0
        for i in range(len(self.layers), num_layers):
            k += 1
            in_feature_dim = inner_feature_dim
            if k == 3:
                inner_feature_dim *= 2
                k = 0
            self.layers.append(gc(in_feature_dim, inner_feature_dim, adj, sparse=sparse, dropout=dropout))
        self.bn = nn.BatchNorm1d(feature_dim * num_nodes)
        for layer_idx, layer in enumerate(self.layers):
            setattr(self, f"gc{layer_idx + 1}", layer)
        if without_fc:
            self.fc = None
        else:
            self.fc = nn.Linear(inner_feature_dim, num_classes)
            nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_classes))

    def forward(self, x):
        batch_size, feature_dim, num_nodes = x.size()
        x = torch.flatten(x, start_dim=1)
        x = self.bn(x)                                   # (input BatchNorm: a torch op, as data_bn of the skeleton model)
        x = torch.reshape(x, (batch_size, feature_dim, num_nodes))
        h = x.permute(0, 2, 1)                           # node-major (B, V, F), channels padded to 4
        pad = (-feature_dim) % 4
        h = (F.pad(h, (0, pad)) if pad else h).contiguous()
        for layer in self.layers:
            h = layer(h)
        h = GroupMeanFunction.apply(h)                   # x.mean(-1): mean over the nodes
        if self.fc is not None:
            h = LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)
        return h
