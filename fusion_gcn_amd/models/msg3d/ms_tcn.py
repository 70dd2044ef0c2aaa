"""Multi-branch temporal convolution of MS-G3D (reference torch_src/models/msg3d/ms_tcn.py:15-109).

Six branches over the same input, concatenated on the channel axis, + a residual, + ReLU:
  four:  1x1 conv -> BN -> ReLU -> (3 x 1) conv with dilation 1..4 and the block's stride -> BN
  one:   1x1 conv -> BN -> ReLU -> (3 x 1) max pooling with the block's stride -> BN
  one:   1x1 conv with the block's stride -> BN
The module tree (``branches.<i>.<j>``, ``residual.{conv,bn}``) is the reference's, so checkpoints load; the sub-modules hold
parameters only.  On the MI355X the five leading 1x1 convolutions that share the un-strided input are ONE row GEMM into a
(B, T, V, 5 * branch) tensor with ONE fused BatchNorm + ReLU pass (BatchNorm is per channel, so the five BatchNorms are one
over the concatenated channels); the dilated convolutions read their channel window of it (row GEMM with a temporal map
ti = to*stride + j*dilation - dilation), the pooling branch its window through fgcn_tmaxpool3."""
import torch
import torch.nn as nn

from ... import fops
from .activation import activation_factory, is_relu


def temporal_map(kernel_size: int, stride: int, dilation: int):
    """ti = to*stride + j*dilation - pad with the reference's 'same' padding (ms_tcn.py:18)."""
    pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
    return (kernel_size, stride, dilation, -pad, 1)


def out_frames(T: int, stride: int) -> int:
    return (T - 1) // stride + 1


class _JoinedBatchNorm:
    """Several per-channel BatchNorms applied to the channel-wise concatenation of their inputs as one: gamma / beta are packed
    forms of the members' parameters (fops.joined_vector: no cat, gradients come back as slices).  The running statistics live
    concatenated in ``cache`` (kept by the owning module): they are gathered from the members only when a member's buffer was
    replaced or written by someone else since the last hand-back (load_state_dict, .to(): address / version stamps) -- two cat
    launches less per block and step -- and the updated ones are written back with one multi-tensor copy."""

    def __init__(self, bns, forms, cache: dict):
        self.bns = list(bns)
        self.training = self.bns[0].training
        self.eps, self.momentum = self.bns[0].eps, self.bns[0].momentum
        self.weight = fops.joined_vector(forms, "heads.gamma", [b.weight for b in self.bns])
        self.bias = fops.joined_vector(forms, "heads.beta", [b.bias for b in self.bns])
        self._cache = cache
        if cache.get("stamp") != self._stamp():
            with torch.no_grad():
                cache["mean"] = torch.cat([b.running_mean for b in self.bns])
                cache["var"] = torch.cat([b.running_var for b in self.bns])
            cache["stamp"] = self._stamp()
        self.running_mean, self.running_var = cache["mean"], cache["var"]
        self.num_batches_tracked = 0        # counted on the members below

    def _stamp(self):
        return tuple((b.running_mean.data_ptr(), b.running_mean._version, b.running_var.data_ptr(), b.running_var._version) for b in self.bns)

    def scatter_running_stats(self) -> None:
        if not self.training:
            return
        with torch.no_grad():
            sizes = [b.num_features for b in self.bns]
            torch._foreach_copy_([b.running_mean for b in self.bns] + [b.running_var for b in self.bns],
                                 list(self.running_mean.split(sizes)) + list(self.running_var.split(sizes)))
            self._cache["stamp"] = self._stamp()     # (the members now hold what the concatenation holds)
            for b in self.bns:
                if fops.deferred_batch_counters.active is not None:
                    fops.deferred_batch_counters.active.buffers.append(b.num_batches_tracked)
                else:
                    b.num_batches_tracked += 1


class TemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1):
        super().__init__()
        pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(pad, 0), stride=(stride, 1),
                              dilation=(dilation, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        self._forms = fops.ParamForms()

    def pre_bn(self, x: torch.Tensor, coff: int = 0):
        """conv(x[..., coff:coff + in_channels]) with BatchNorm partial sums -> (y, partials)"""
        k, s, d = self.conv.kernel_size[0], self.conv.stride[0], self.conv.dilation[0]
        return fops.conv_params(x, self._forms, "conv", [self.conv.weight], [self.conv.bias], tmap=temporal_map(k, s, d),
                                T_out=out_frames(x.shape[1], s), stats=self.bn.training, zero_bias_grad=self.bn.training, in_coff=coff)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y, part = self.pre_bn(x)
        return fops.bn_act(y, part, self.bn)


class MultiScale_TemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilations=None, residual=True, residual_kernel_size=1,
                 activation="relu"):
        super().__init__()
        dilations = [1, 2, 3, 4] if dilations is None else dilations
        self.num_branches = len(dilations) + 2
        assert out_channels % self.num_branches == 0, "# out channels should be multiples of # branches"
        bc = out_channels // self.num_branches
        self.stride, self.branch_channels = stride, bc

        def head(**kw):
            return [nn.Conv2d(in_channels, bc, kernel_size=1, padding=0, **kw), nn.BatchNorm2d(bc)]
        self.branches = nn.ModuleList(
            nn.Sequential(*head(), activation_factory(activation), TemporalConv(bc, bc, kernel_size=kernel_size, stride=stride, dilation=d))
            for d in dilations)
        self.branches.append(nn.Sequential(*head(), activation_factory(activation),
                                           nn.MaxPool2d(kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0)), nn.BatchNorm2d(bc)))
        self.branches.append(nn.Sequential(*head(stride=(stride, 1))))
        if not residual:
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self.residual = lambda x: x
        else:
            self.residual = TemporalConv(in_channels, out_channels, kernel_size=residual_kernel_size, stride=stride)
        self.act = activation_factory(activation)
        self._forms = fops.ParamForms()
        self._joined_stats = {}                  # concatenated running statistics of the five heads' BatchNorms (_JoinedBatchNorm)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        bc, s = self.branch_channels, self.stride
        lead = self.branches[:-1]                       # the five branches that start with an un-strided 1x1 conv + BN + act
        relu_heads = is_relu(lead[0][2])
        # one GEMM + one BatchNorm/activation pass for the five heads
        joined = _JoinedBatchNorm([b[1] for b in lead], self._forms, self._joined_stats)
        h, part = fops.conv_params(x, self._forms, "heads", [b[0].weight for b in lead], [b[0].bias for b in lead],      # (1, Cin, 5 bc)
                                   stats=joined.training, zero_bias_grad=joined.training)
        h = fops.bn_act(h, part, joined, relu=relu_heads)
        joined.scatter_running_stats()
        # dilated (3 x 1) convolutions and the max pooling on their channel windows of h, read and differentiated in place
        tconvs = [b[3] for b in lead[:-1]]
        train = tconvs[0].bn.training
        ys, pooled, parts = fops.window_branches_params(
            h, self._forms, "windows", [t.conv for t in tconvs],
            [temporal_map(t.conv.kernel_size[0], t.conv.stride[0], t.conv.dilation[0]) for t in tconvs], bc, s, out_frames(x.shape[1], s),
            stats=train, zero_bias_grad=train)
        # the six branches' BatchNorms write their channel windows of one tensor (fops.bn_cat: no torch.cat, and the backward reads
        # the windows of its gradient in place)
        same = len({t.bn.eps for t in tconvs} | {lead[-1][4].eps, self.branches[-1][1].eps}) == 1 and bc % 4 == 0
        last = self.branches[-1]
        y_last, part_last = fops.conv_params(x, self._forms, "strided", [last[0].weight], [last[0].bias], tmap=(1, s, 0, 0, 1),
                                             T_out=out_frames(x.shape[1], s), stats=last[1].training, zero_bias_grad=last[1].training)
        pooled_part = fops.col_stats(pooled) if lead[-1][4].training else pooled.new_empty(0)
        branches = [(y, part, t.bn) for y, part, t in zip(ys, parts, tconvs)] + [(pooled, pooled_part, lead[-1][4]), (y_last, part_last, last[1])]
        if same:
            out = fops.bn_cat(branches)
        else:
            out = torch.cat([fops.bn_act(y, part, bn) for y, part, bn in branches], dim=-1)
        res = self.residual(x)
        relu_out = is_relu(self.act)
        if isinstance(res, torch.Tensor):
            return fops.add_act(out, res, relu=relu_out)
        return fops.add_act(out, torch.zeros_like(out), relu=True) if relu_out else out
