"""``model: msg3d`` -- MS-G3D (Liu et al., CVPR 2020) on the MI355X kernels; drop-in for the reference's
torch_src/models/msg3d/msg3d.py (MS_G3D :19-73, MultiWindow_MS_G3D :76-110, Model :113-182).

Same constructor (``Model(data_shape={"skeleton": (M, T, V, C)}, num_classes, graph, num_gcn_scales=13, num_g3d_scales=6)``), module
tree and state-dict keys (``gcn3d<i>.gcn3d.<w>.{gcn3d.1.{A_res,mlp.layers.*},out_conv,out_bn}``, ``sgcn<i>.{0,1,2}.*``, ``tcn<i>.*``,
``data_bn``, ``fc``) and, from the same seed, the same initial parameters, so checkpoints are interchangeable.  Activations travel
channels-last (B, T, V, C) between the blocks (the reference's input (N, M, T, V, C) already is); the arithmetic of every block
runs in libfgcn through the differentiable ops of fusion_gcn_amd/fops.py -- row GEMMs with BatchNorm partial sums in the
epilogue, fused BatchNorm / residual / activation passes, the feature-major node aggregation, temporal max pooling and window
unfolding -- and raises without the built library / off gfx950.  ``data_bn`` (BatchNorm1d on the 3-channel input) and the
loss are torch-ROCm ops, as in the AGCN model.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import fops
from ...block import GroupMeanFunction, LinearFunction, data_bn
from .mlp import MLP
from .ms_gcn import MultiScale_GraphConv as MS_GCN
from .ms_gtcn import SpatialTemporal_MS_GCN, UnfoldTemporalWindows
from .ms_tcn import MultiScale_TemporalConv as MS_TCN

INPUT_CHANNELS = 3        # the reference singles the first block out by its 3 input channels (msg3d.py:38-39)


class MS_G3D(nn.Module):
    """One G3D pathway: unfold the temporal windows, aggregate over the spatial-temporal graph, collapse the window axis with a
    (1, window, 1) Conv3d, BatchNorm, no activation."""

    def __init__(self, in_channels, out_channels, A_binary, num_scales, window_size, window_stride, window_dilation, embed_factor=1,
                 activation="relu"):
        super().__init__()
        self.window_size, self.out_channels = window_size, out_channels
        if embed_factor == 1:
            self.in1x1 = nn.Identity()
            self.embed_channels_in = self.embed_channels_out = in_channels
            if in_channels == INPUT_CHANNELS:          # the first block changes channels right away, the others at the collapse
                self.embed_channels_out = out_channels
        else:
            self.embed_channels_in = self.embed_channels_out = out_channels // embed_factor
            self.in1x1 = MLP(in_channels, [self.embed_channels_in])
        self.gcn3d = nn.Sequential(
            UnfoldTemporalWindows(window_size, window_stride, window_dilation),
            SpatialTemporal_MS_GCN(in_channels=self.embed_channels_in, out_channels=self.embed_channels_out, A_binary=A_binary,
                                   num_scales=num_scales, window_size=window_size, use_Ares=True))
        self.out_conv = nn.Conv3d(self.embed_channels_out, out_channels, kernel_size=(1, window_size, 1))
        self.out_bn = nn.BatchNorm2d(out_channels)
        self._forms = fops.ParamForms()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, V, _ = x.shape
        h = self.gcn3d(self.in1x1(x))                                   # (B, T', window * V, embed_out)
        Tw, ws = h.shape[1], self.window_size
        # collapse: out[(b, t', v)] = sum_j W_j . h[(b, t', j*V + v)] -- a temporal conv over the frame index t'*window + j with
        # `window` taps and stride `window`
        h = h.view(B, Tw * ws, V, h.shape[-1])
        y, part = fops.conv_params(h, self._forms, "out_conv", [self.out_conv.weight], [self.out_conv.bias], tmap=(ws, ws, 1, 0, 1),
                                   T_out=Tw, stats=self.out_bn.training, zero_bias_grad=self.out_bn.training)   # (window, embed_out, out)
        return fops.bn_act(y, part, self.out_bn)


class MultiWindow_MS_G3D(nn.Module):
    def __init__(self, in_channels, out_channels, A_binary, num_scales, window_sizes=None, window_stride=1, window_dilations=None):
        super().__init__()
        window_sizes = [3, 5] if window_sizes is None else window_sizes
        window_dilations = [1, 1] if window_dilations is None else window_dilations
        self.gcn3d = nn.ModuleList(MS_G3D(in_channels, out_channels, A_binary, num_scales, size, window_stride, dilation)
                                   for size, dilation in zip(window_sizes, window_dilations))

    def forward(self, x: torch.Tensor):
        out = None
        for pathway in self.gcn3d:
            y = pathway(x)
            out = y if out is None else fops.add_act(out, y, relu=False)
        return out


# (G3D stride, output width) of the three stages; every stage is  relu(sgcn(x) + gcn3d(x)) -> tcn
_STAGES = ((1, 96), (2, 192), (2, 384))


class Model(nn.Module):
    def __init__(self, data_shape, num_classes, graph, **kwargs):
        super().__init__()
        num_persons, _, num_joints, num_channels = data_shape["skeleton"]      # (persons, frames, joints, channels)
        num_gcn_scales, num_g3d_scales = kwargs.get("num_gcn_scales", 13), kwargs.get("num_g3d_scales", 6)
        A_binary = graph.get_adjacency_matrix().astype(np.float64)
        self.data_bn = nn.BatchNorm1d(num_persons * num_channels * num_joints)
        cin = INPUT_CHANNELS
        for i, (stride, cout) in enumerate(_STAGES, start=1):
            setattr(self, f"gcn3d{i}", MultiWindow_MS_G3D(cin, cout, A_binary, num_g3d_scales, window_stride=stride))
            sgcn = nn.Sequential(MS_GCN(num_gcn_scales, cin, cin if i > 1 else cout, A_binary, disentangled_agg=True),
                                 MS_TCN(cin if i > 1 else cout, cout, stride=stride), MS_TCN(cout, cout))
            sgcn[-1].act = nn.Identity()
            setattr(self, f"sgcn{i}", sgcn)
            setattr(self, f"tcn{i}", MS_TCN(cout, cout))
            cin = cout
        self.fc = nn.Linear(cin, num_classes)

    def mark_packed_stale(self) -> None:
        """Force the re-pack of every weight form at the next forward (what an optimizer step does through the version counters)."""
        fops.mark_forms_stale(self)

    def prepare_recording(self) -> None:
        """GraphStep hook: the one-launch re-pack plan is built (device tables, an H2D copy) before the step is recorded."""
        fops.mark_forms_stale(self)
        fops.refresh_forms(self)

    def recording_pins(self) -> list:
        """GraphStep hook: the forms and the re-pack plan a recording made now replays."""
        return fops.recording_pins(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        N, M, T, V, C = x.size()
        fops.refresh_forms(self)
        h = data_bn(x, self.data_bn)          # BatchNorm1d over (m, v, c) -> channels-last (B, T, V, 4): 3 channels travel as 4 (4th zero)
        with fops.deferred_batch_counters(), fops.zero_pool(self):                    # one add for all BatchNorm counters, one fill for all zero bias gradients
            for i in range(1, len(_STAGES) + 1):
                s = getattr(self, f"sgcn{i}")(h)
                g = getattr(self, f"gcn3d{i}")(h)
                h = getattr(self, f"tcn{i}")(fops.add_act(s, g, relu=True))
        h = GroupMeanFunction.apply(h.reshape(N, -1, h.size(-1)))                     # mean over persons, frames and joints
        return LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)
