"""AGCN model (2s-AGCN, Shi et al. CVPR 2019) on the MI355X HIP kernels — drop-in for the reference's
``torch_src/models/mmargcn/agcn.py`` (classes TemporalConv :37-51, SpatialGraphConv :54-115,
SpatialTemporalConv :118-136, Model :139-200).

Same constructor signatures, attribute / state-dict names (``l0..l9``, ``gcn1.adj_a`` buffer, ``gcn1.adj_b``,
``conv_a/b/d.{0,1,2}``, ``down.{0,1}``, ``tcn1.{conv,bn}``, ``residual.{conv,bn}``, ``data_bn``, ``fc``), init
distributions and train / eval BatchNorm semantics, so checkpoints are interchangeable.  The nn.Conv2d /
nn.BatchNorm2d sub-modules are *parameter containers only*: their forward is never called.  The arithmetic of
every block runs in libfgcn (fusion_gcn_amd/block.py); without the built library, or off gfx950, forward raises.

``Model.forward`` takes the reference's input layout (N, M, T, V, C) float32.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...block import BlockConfig, GroupMeanFunction, LinearFunction, STBlockFunction, bn_names, pack_weights, param_names
from ...util.partition_strategy import GraphPartitionStrategy


def conv_branch_init(conv, branches):
    weight = conv.weight
    n, k1, k2 = weight.size(0), weight.size(1), weight.size(2)
    nn.init.normal_(weight, 0, math.sqrt(2. / (n * k1 * k2 * branches)))
    nn.init.constant_(conv.bias, 0)


def conv_init(conv):
    nn.init.kaiming_normal_(conv.weight, mode="fan_out")
    nn.init.constant_(conv.bias, 0)


def bn_init(bn, scale):
    nn.init.constant_(bn.weight, scale)
    nn.init.constant_(bn.bias, 0)


class _KernelBacked(nn.Module):
    """Sub-modules of a block hold parameters only; the block's fused kernels do the arithmetic."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} is evaluated inside SpatialTemporalConv's HIP kernels; "
                           "call the enclosing block (or Model) instead")


class TemporalConv(_KernelBacked):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 9, stride: int = 1):
        super().__init__()
        pad = int((kernel_size - 1) / 2)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(pad, 0),
                              stride=(stride, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        conv_init(self.conv)
        bn_init(self.bn, 1)


class SpatialGraphConv(_KernelBacked):
    def __init__(self, in_channels: int, out_channels: int, adj: np.ndarray, coff_embedding: int = 4,
                 num_subsets: int = 3):
        super().__init__()
        if coff_embedding != 4 or num_subsets != 3:
            raise ValueError("the HIP AGCN block implements coff_embedding=4, num_subsets=3 (the reference's only use)")
        inter_channels = out_channels // coff_embedding
        self.inter_channels = inter_channels
        self.num_subsets = num_subsets

        self.adj_b = nn.Parameter(torch.from_numpy(adj.astype(np.float32)))
        nn.init.constant_(self.adj_b, 1e-6)
        self.register_buffer("adj_a", torch.from_numpy(adj.astype(np.float32)))
        self.adj_c = [None] * self.num_subsets      # last forward's data-dependent adjacencies (metrics side output)

        self.conv_a = nn.ModuleList()
        self.conv_b = nn.ModuleList()
        self.conv_d = nn.ModuleList()
        for _ in range(self.num_subsets):
            self.conv_a.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_b.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_d.append(nn.Conv2d(in_channels, out_channels, 1))

        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = lambda x: x

        self.bn = nn.BatchNorm2d(out_channels)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for i in range(self.num_subsets):
            conv_branch_init(self.conv_d[i], self.num_subsets)


class SpatialTemporalConv(nn.Module):
    """One AGCN block.  ``forward`` takes and returns channels-last (B, T, V, C) tensors (the model keeps that
    layout between blocks); ``forward_nchw`` accepts the reference's (B, C, T, V)."""

    # canonical (block.py) tensor name -> attribute path; subclasses with other spellings override
    _ADJ_PARAM = "gcn1.adj_b"

    def __init__(self, in_channels, out_channels, adj, stride=1, residual=True, static_adjacency: bool = False,
                 fused_spatial: bool = True):
        super().__init__()
        self.gcn1 = SpatialGraphConv(in_channels, out_channels, adj)
        self.tcn1 = TemporalConv(out_channels, out_channels, stride=stride)
        self.out_channels = out_channels
        if not residual:
            self.residual = lambda x: 0
            res = "none"
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = lambda x: x
            res = "identity"
        else:
            self.residual = TemporalConv(in_channels, out_channels, kernel_size=1, stride=stride)
            res = "conv"
        self.cfg = BlockConfig(cin=in_channels, cout=out_channels, stride=stride, residual=res,
                               has_down=in_channels != out_channels, static_adjacency=static_adjacency,
                               fused_spatial=fused_spatial)
        self.cfg.validate()
        self._wcache = None

    # -- tensors by canonical name ------------------------------------------------------------------------------------
    def _tensor(self, name: str) -> torch.Tensor:
        if name == "gcn1.adj_b":
            name = self._ADJ_PARAM
        obj = self
        for part in name.split("."):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        return obj

    def _adj_a(self) -> torch.Tensor:
        return self.gcn1.adj_a

    def _block_buffers(self) -> Dict[str, torch.Tensor]:
        bufs = {"gcn1.adj_a": self._adj_a()}
        for bn in bn_names(self.cfg):
            mod = self._tensor(bn)
            bufs[f"{bn}.running_mean"], bufs[f"{bn}.running_var"] = mod.running_mean, mod.running_var
        return bufs

    _defer_nbt = False   # a Model bumps the num_batches_tracked counters of all its blocks with one multi-tensor add

    def nbt_buffers(self):
        return [self._tensor(bn).num_batches_tracked for bn in bn_names(self.cfg)]

    def _packed(self, params):
        key = (ops.get_math_mode(),) + tuple((p.data_ptr(), p._version) for p in params)   # the packed forms depend on the mode
        if self._wcache is None or self._wcache[0] != key:
            P = dict(zip(param_names(self.cfg), params))
            self._wcache = (key, pack_weights(P, self.cfg))
        return self._wcache[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        names = param_names(self.cfg)
        params = [self._tensor(n) for n in names]
        W = self._packed(params)
        holder = {}
        out = STBlockFunction.apply(x, self.cfg, self.training, self._block_buffers(), W, holder, *params)
        if self.training and not self._defer_nbt:
            torch._foreach_add_(self.nbt_buffers(), 1)
        c = holder.get("adj_c")
        self.gcn1.adj_c = [c[:, k] for k in range(3)] if c is not None else [None] * 3
        return out

    def forward_nchw(self, x: torch.Tensor) -> torch.Tensor:
        xc = x.permute(0, 2, 3, 1)
        pad = self.cfg.cx - self.cfg.cin
        xc = (F.pad(xc, (0, pad)) if pad else xc).contiguous()
        return self.forward(xc).permute(0, 3, 1, 2)


class Model(nn.Module):
    _BLOCK = SpatialTemporalConv

    def __init__(self, data_shape: tuple, num_classes: int, graph, num_layers: int = 10, start_feature_size: int = 64,
                 without_fc=False, dropout: float = 0., static_adjacency: bool = False, fused_spatial: bool = True,
                 adjacency_matrix: Optional[np.ndarray] = None):
        super().__init__()
        # data_shape = (num_persons, num_frames, num_joints, num_channels)
        num_persons, _, num_joints, num_channels = data_shape
        adj = adjacency_matrix if adjacency_matrix is not None else GraphPartitionStrategy().get_adjacency_matrix_array(graph)
        self.data_bn = nn.BatchNorm1d(num_persons * num_channels * num_joints)
        f = start_feature_size
        kw = dict(static_adjacency=static_adjacency, fused_spatial=fused_spatial)
        self.layers = [
            self._BLOCK(num_channels, f, adj, residual=False, **kw),
            self._BLOCK(f, f, adj, **kw),
            self._BLOCK(f, f, adj, **kw),
            self._BLOCK(f, f, adj, **kw),
            self._BLOCK(f, f * 2, adj, stride=2, **kw),
            self._BLOCK(f * 2, f * 2, adj, **kw),
            self._BLOCK(f * 2, f * 2, adj, **kw),
            self._BLOCK(f * 2, f * 4, adj, stride=2, **kw),
            self._BLOCK(f * 4, f * 4, adj, **kw),
            self._BLOCK(f * 4, f * 4, adj, **kw),
        ]
        self.layers = self.layers[:min(len(self.layers), num_layers)]
        if dropout > 0:   # reference: a Dropout after every block but the last, shifting the l<i> indices
            for i in range(1, len(self.layers) * 2 - 1, 2):
                self.layers.insert(i, nn.Dropout(dropout, inplace=True))
        for layer_idx, layer in enumerate(self.layers):
            setattr(self, f"l{layer_idx}", layer)
        last = [l for l in self.layers if isinstance(l, SpatialTemporalConv)][-1]
        if without_fc:
            self.fc = None
            self.out_channels = last.out_channels
        else:
            self.fc = nn.Linear(last.out_channels, num_classes)
            nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_classes))
            self.out_channels = num_classes
        bn_init(self.data_bn, 1)

    def _blocks_input(self, x: torch.Tensor) -> torch.Tensor:
        """(N, M, T, V, C) -> data_bn over the (m, v, c) channels with statistics across (n, t)
        -> channels-last (N*M, T, V, C padded to 4)."""
        N, M, T, V, C = x.size()
        h = x.permute(0, 1, 3, 4, 2).contiguous().view(N, M * V * C, T)
        h = self.data_bn(h)
        h = h.view(N, M, V, C, T).permute(0, 1, 4, 2, 3).reshape(N * M, T, V, C)
        pad = (-C) % 4
        if pad:
            h = F.pad(h, (0, pad))
        return h.contiguous()

    def _bump_batch_counters(self) -> None:
        """num_batches_tracked += 1 for every block BatchNorm in one launch (26 scalar adds otherwise)."""
        blocks = [m for m in self.modules() if isinstance(m, SpatialTemporalConv)]
        for b in blocks:
            b._defer_nbt = True
        if self.training:
            torch._foreach_add_([t for b in blocks for t in b.nbt_buffers()], 1)

    def forward(self, x):
        N, M, T, V, C = x.size()
        h = self._blocks_input(x)
        self._bump_batch_counters()
        for layer in self.layers:
            h = layer(h)
        # (N*M, T', V, C') -> mean over (T', V) then over persons
        c_new = h.size(-1)
        h = GroupMeanFunction.apply(h.view(N, -1, c_new))   # mean over persons, frames and joints (equal-sized groups)
        if self.fc is not None:    # nn.Linear is the parameter container; the arithmetic is the row GEMM
            h = LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)
        return h
