"""Spatial-temporal multi-scale graph convolution over unfolded temporal windows -- the "G3D" pathway (reference
torch_src/models/msg3d/ms_gtcn.py:24-126).

``UnfoldTemporalWindows`` turns the ``window`` frames around every stride-th frame into ``window * V`` nodes of one graph
(fgcn_unfold_windows); ``SpatialTemporal_MS_GCN`` aggregates over the (num_scales * window*V, window*V) stack of normalised k-hop
adjacencies of the tiled skeleton graph + a learnable residual, then a linear 1x1 MLP, a residual and the activation.  With up to
135 nodes this is beyond the <= 32-joint register kernels of the AGCN block: the aggregation is fops.node_mix (feature-major
row GEMM with the stacked matrix as the shared weight)."""
import numpy as np
import torch
import torch.nn as nn

from ... import fops
from .activation import activation_factory, is_relu
from .mlp import MLP
from .ms_gcn import k_hop_stack


class UnfoldTemporalWindows(nn.Module):
    def __init__(self, window_size, window_stride, window_dilation=1):
        super().__init__()
        self.window_size, self.window_stride, self.window_dilation = window_size, window_stride, window_dilation
        self.padding = (window_size + (window_size - 1) * (window_dilation - 1) - 1) // 2

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, T, V, C) -> (B, T', window * V, C)"""
        return fops.unfold_windows(x, self.window_size, self.window_stride, self.window_dilation)


class SpatialTemporal_MS_GCN(nn.Module):
    def __init__(self, in_channels, out_channels, A_binary, num_scales, window_size, disentangled_agg=True, use_Ares=True,
                 residual=False, dropout=0, activation="relu"):
        super().__init__()
        if not (disentangled_agg and use_Ares) or residual:
            raise NotImplementedError("the HIP G3D block implements the reference model's setting: disentangled aggregation, learnable "
                                      "residual adjacency, no block residual")
        self.num_scales, self.window_size, self.use_Ares, self.in_channels = num_scales, window_size, use_Ares, in_channels
        self.A_scales = torch.from_numpy(k_hop_stack(self.build_spatial_temporal_graph(A_binary, window_size), num_scales))
        self.V = len(A_binary)
        # (the reference draws a normal tensor first and then overwrites it uniformly: both draws are kept so that the same seed
        # leaves the generator -- and every parameter initialised after this one -- in the same state)
        self.A_res = nn.init.uniform_(nn.Parameter(torch.randn(self.A_scales.shape)), -1e-6, 1e-6)
        self.mlp = MLP(in_channels * num_scales, [out_channels], dropout=dropout, activation="linear")
        self.residual = lambda x: 0
        self.act = activation_factory(activation)
        self._forms = fops.ParamForms()

    @staticmethod
    def build_spatial_temporal_graph(A_binary: np.ndarray, window_size: int) -> np.ndarray:
        """every frame of the window fully connected to every other through the skeleton edges + self loops (reference :101-109)"""
        return np.tile(A_binary + np.eye(len(A_binary), dtype=A_binary.dtype), (window_size, window_size)).copy()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.A_scales = fn(self.A_scales)
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, T', window * V, C) -> (B, T', window * V, out)"""
        if self.A_scales.device != x.device:
            self.A_scales = self.A_scales.to(x.device)
        agg = fops.node_mix_params(x, self._forms, "A", self.A_scales, self.A_res, self.num_scales)
        conv, bn, _ = self.mlp.layers
        y, part = fops.conv_params(agg, self._forms, "mlp", [conv.weight], [conv.bias], stats=bn.training, zero_bias_grad=bn.training,
                                   scales=self.num_scales)
        return fops.bn_act(y, part, bn, relu=is_relu(self.act))            # linear MLP, no residual, then the block's activation
