"""Point-wise MLP of the MS-G3D blocks (reference torch_src/models/msg3d/mlp.py:14-30): per layer Conv2d 1x1 -> BatchNorm2d ->
activation, registered as ``layers.{0,1,2}`` (+3 per further layer).  The sub-modules hold parameters; the arithmetic is the row
GEMM with BatchNorm partial sums in its epilogue + the fused BatchNorm / activation kernel (fops.conv_rows, fops.bn_act)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import fops
from .activation import activation_factory, is_relu


def pointwise_weight(conv: nn.Module, k_pad: int = 0) -> torch.Tensor:
    """(O, I, 1, 1[, 1]) -> the packed (1, I + k_pad, O) matrix of the row GEMM (differentiable re-layout of a small tensor)."""
    w = conv.weight.reshape(conv.weight.shape[0], -1).t()
    return (F.pad(w, (0, 0, 0, k_pad)) if k_pad else w).unsqueeze(0)


class MLP(nn.Module):
    def __init__(self, in_channels, out_channels, activation="relu", dropout=0):
        super().__init__()
        if dropout > 0.001:
            raise NotImplementedError("dropout inside the HIP MLP is not built (the reference's MS-G3D model uses 0)")
        self.layers = nn.ModuleList()
        for cin, cout in zip([in_channels] + list(out_channels), out_channels):
            self.layers += [nn.Conv2d(cin, cout, kernel_size=1), nn.BatchNorm2d(cout), activation_factory(activation)]

    def forward(self, x: torch.Tensor, weights=None) -> torch.Tensor:
        """x (B, T, V, C) channels-last; ``weights``: per layer an already packed (1, K, N) matrix (callers whose input channels are
        laid out differently from the reference's pass their own re-layout of layers[3i].weight)."""
        for i in range(0, len(self.layers), 3):
            conv, bn, act = self.layers[i:i + 3]
            w = weights[i // 3] if weights is not None else pointwise_weight(conv, x.shape[-1] - conv.in_channels)
            y, part = fops.conv_rows(x, w, conv.bias, stats=bn.training, zero_bias_grad=bn.training)
            x = fops.bn_act(y, part, bn, relu=is_relu(act))
        return x
