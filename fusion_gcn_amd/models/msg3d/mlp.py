"""Point-wise MLP of the MS-G3D blocks (reference torch_src/models/msg3d/mlp.py:14-30): per layer Conv2d 1x1 -> BatchNorm2d ->
activation, registered as ``layers.{0,1,2}`` (+3 per further layer).  The sub-modules hold parameters; the arithmetic is the row
GEMM with BatchNorm partial sums in its epilogue + the fused BatchNorm / activation kernel (fops.conv_params, fops.bn_act); the
packed forms of the weights live in ``_forms`` and are refreshed with the whole model's (fops.refresh_forms)."""
import torch
import torch.nn as nn

from ... import fops
from .activation import activation_factory, is_relu


class MLP(nn.Module):
    def __init__(self, in_channels, out_channels, activation="relu", dropout=0):
        super().__init__()
        if dropout > 0.001:
            raise NotImplementedError("dropout inside the HIP MLP is not built (the reference's MS-G3D model uses 0)")
        self.layers = nn.ModuleList()
        for cin, cout in zip([in_channels] + list(out_channels), out_channels):
            self.layers += [nn.Conv2d(cin, cout, kernel_size=1), nn.BatchNorm2d(cout), activation_factory(activation)]
        self._forms = fops.ParamForms()

    def forward(self, x: torch.Tensor, scales: int = 1) -> torch.Tensor:
        """x (B, T, V, C) channels-last.  ``scales`` > 1: x is a multi-scale aggregate (fops.node_mix) whose channel is s * width + c;
        the first layer's weight (O, scales * C) is read scale-major, `width - C` zero pad channels per scale skipped."""
        for i in range(0, len(self.layers), 3):
            conv, bn, act = self.layers[i:i + 3]
            y, part = fops.conv_params(x, self._forms, f"layers.{i}", [conv.weight], [conv.bias], stats=bn.training,
                                       zero_bias_grad=bn.training, scales=scales if i == 0 else 1)
            x = fops.bn_act(y, part, bn, relu=is_relu(act))
        return x
