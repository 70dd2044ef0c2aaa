"""Multi-scale (disentangled k-hop) graph convolution (reference torch_src/models/msg3d/ms_gcn.py:24-64).

``support = einsum('vu,nctu->nctv', A_powers + A_res, x)`` over the 13 stacked normalised k-hop adjacencies, the scales moved
into the channel axis, then a 1x1 MLP.  Here: fops.node_mix (the stacked matrix as the shared weight of a row GEMM over the
feature-major image) writes the (B, T, V, S*C) aggregate directly in the channel order the MLP's weight expects (s*C + c)."""
import numpy as np
import torch
import torch.nn as nn

from ... import fops
from ...util.graph import get_k_adjacency
from .mlp import MLP


def normalize_adjacency_matrix(a: np.ndarray) -> np.ndarray:
    """D^-1/2 A D^-1/2 in float32 (reference :17-21)."""
    scale = np.power(a.sum(-1), -0.5)
    return (scale[:, None] * a * scale[None, :]).astype(np.float32)


def k_hop_stack(a: np.ndarray, num_scales: int) -> np.ndarray:
    """(num_scales * V, V): normalised exact-k-hop adjacencies with self loops, k = 0 .. num_scales-1, stacked along the rows."""
    return np.concatenate([normalize_adjacency_matrix(get_k_adjacency(a, k, with_self=True)) for k in range(num_scales)])


class MultiScale_GraphConv(nn.Module):
    def __init__(self, num_scales, in_channels, out_channels, A_binary, disentangled_agg=True, use_mask=True, dropout=0,
                 activation="relu"):
        super().__init__()
        if not (disentangled_agg and use_mask):
            raise NotImplementedError("the HIP MS-GCN implements the reference model's setting: disentangled aggregation with the "
                                      "learnable residual mask")
        self.num_scales, self.in_channels = num_scales, in_channels
        self.A_powers = torch.from_numpy(k_hop_stack(A_binary, num_scales))           # plain attribute, as in the reference
        self.use_mask = use_mask
        self.A_res = nn.init.uniform_(nn.Parameter(torch.empty(self.A_powers.shape)), -1e-6, 1e-6)
        self.mlp = MLP(in_channels * num_scales, [out_channels], dropout=dropout, activation=activation)
        self._forms = fops.ParamForms()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.A_powers = fn(self.A_powers)
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.A_powers.device != x.device:
            self.A_powers = self.A_powers.to(x.device)
        agg = fops.node_mix_params(x, self._forms, "A", self.A_powers, self.A_res, self.num_scales)       # (B, T, V, S * C)
        return self.mlp(agg, scales=self.num_scales)
