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
from ...block import (BlockConfig, GroupMeanFunction, LinearFunction, STBlockFunction, bn_names, data_bn, pack_weights, param_names,
                      pool_epilogue_ok, zero_bias_floats)
from ...util.partition_strategy import GraphPartitionStrategy


# ---- initialisation (reference agcn.py:18-34) -------------------------------------------------------------------------------
# Every convolution of the network is drawn from N(0, 2 / fan) with a bias of zero; what differs is the fan: the kaiming
# fan-out (out_channels x kernel area) everywhere, except conv_d, whose three subset branches are summed and therefore share
# one fan (out x in x kernel height x branches).  BatchNorms are constants.  The helpers keep the reference's names.
def _gaussian_conv(conv, fan: float) -> None:
    with torch.no_grad():
        conv.weight.normal_(0, math.sqrt(2.0) / math.sqrt(fan))
        conv.bias.zero_()


def conv_init(conv):
    """kaiming-normal, fan-out mode: fan = out_channels x kernel positions (Conv2d and the 1-D graph models' Conv1d alike)."""
    _gaussian_conv(conv, conv.weight.shape[0] * math.prod(conv.weight.shape[2:]))


def conv_branch_init(conv, branches):
    out_c, in_c, k = conv.weight.shape[:3]
    _gaussian_conv(conv, out_c * in_c * k * branches)


def bn_init(bn, scale):
    with torch.no_grad():
        bn.weight.fill_(scale)
        bn.bias.zero_()


def _identity(x):
    return x


def _nothing(x):
    return 0


class _KernelBacked(nn.Module):
    """Sub-modules of a block hold parameters only; the block's fused kernels do the arithmetic."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} is evaluated inside SpatialTemporalConv's HIP kernels; "
                           "call the enclosing block (or Model) instead")


class TemporalConv(_KernelBacked):
    """Parameters of BN(Conv2d((k, 1), stride (s, 1), 'same' padding in time)) -- no activation (reference :37-51)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 9, stride: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, (kernel_size, 1), stride=(stride, 1), padding=((kernel_size - 1) // 2, 0))
        self.bn = nn.BatchNorm2d(out_channels)
        conv_init(self.conv)
        bn_init(self.bn, 1)


class GraphConvParams(_KernelBacked):
    """Parameters of the adaptive graph convolution, shared by both spellings of the model: the learned adjacency (``adj_b`` here,
    ``PA`` in models/agcn), the three (theta, phi, W_d) 1x1 convolutions per subset, the optional channel-matching branch and
    the output BatchNorm.  Registration order = the reference's state-dict order (:62-83); initial values = :62-63,86-94."""

    ADJ_PARAM = "adj_b"
    BRANCHES = ("conv_a", "conv_b", "conv_d")        # theta, phi (out/4 channels each), W_d (out channels)

    def __init__(self, in_channels: int, out_channels: int, adjacency: torch.Tensor, num_subsets: int):
        super().__init__()
        self.setattr_adjacency(adjacency)
        self.adj_c = [None] * num_subsets            # last forward's data-dependent adjacencies (metrics side output)
        inner = self.embedding_channels = out_channels // 4
        for name in self.BRANCHES:
            width = out_channels if name == "conv_d" else inner
            setattr(self, name, nn.ModuleList(nn.Conv2d(in_channels, width, 1) for _ in range(num_subsets)))
        self.down = (nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
                     if in_channels != out_channels else _identity)
        self.bn = nn.BatchNorm2d(out_channels)
        self.reset_parameters(num_subsets)

    def setattr_adjacency(self, adjacency: torch.Tensor) -> None:
        setattr(self, self.ADJ_PARAM, nn.Parameter(torch.full_like(adjacency, 1e-6)))
        self.register_buffer("adj_a", adjacency.clone())

    def reset_parameters(self, num_subsets: int) -> None:
        """One pass over the sub-modules in registration order (= the order the reference draws in), then the two overrides:
        the output BatchNorm starts at 1e-6 (the block starts as its shortcut) and conv_d takes the branch fan."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for conv in self.conv_d:
            conv_branch_init(conv, num_subsets)


class SpatialGraphConv(GraphConvParams):
    def __init__(self, in_channels: int, out_channels: int, adj: np.ndarray, coff_embedding: int = 4,
                 num_subsets: int = 3):
        if coff_embedding != 4 or num_subsets != 3:
            raise ValueError("the HIP AGCN block implements coff_embedding=4, num_subsets=3 (the reference's only use)")
        super().__init__(in_channels, out_channels, torch.from_numpy(np.asarray(adj, dtype=np.float32)), num_subsets)
        self.inter_channels, self.num_subsets = self.embedding_channels, num_subsets


def residual_kind(in_channels: int, out_channels: int, stride: int, residual: bool) -> str:
    """Which shortcut a block takes around gcn + tcn (reference :125-132): none / the input itself / a strided 1x1 TemporalConv."""
    if not residual:
        return "none"
    return "identity" if (in_channels == out_channels and stride == 1) else "conv"


class SpatialTemporalConv(nn.Module):
    """One AGCN block.  ``forward`` takes and returns channels-last (B, T, V, C) tensors (the model keeps that
    layout between blocks); ``forward_nchw`` accepts the reference's (B, C, T, V)."""

    # canonical (block.py) tensor name -> attribute path; subclasses with other spellings override
    _ADJ_PARAM = "gcn1.adj_b"

    _GCN, _TCN = SpatialGraphConv, TemporalConv

    def __init__(self, in_channels, out_channels, adj, stride=1, residual=True, static_adjacency: bool = False,
                 fused_spatial: bool = True):
        super().__init__()
        self.gcn1 = self._GCN(in_channels, out_channels, adj)
        self.tcn1 = self._TCN(out_channels, out_channels, stride=stride)
        self.out_channels = out_channels
        kind = residual_kind(in_channels, out_channels, stride, residual)
        self.residual = {"none": lambda: _nothing, "identity": lambda: _identity,
                         "conv": lambda: self._TCN(in_channels, out_channels, kernel_size=1, stride=stride)}[kind]()
        self.cfg = BlockConfig(cin=in_channels, cout=out_channels, stride=stride, residual=kind,
                               has_down=in_channels != out_channels, static_adjacency=static_adjacency,
                               fused_spatial=fused_spatial)
        self.cfg.validate()
        self._wcache = None          # ((math mode, parameter addresses), packing.PackedWeights)
        self._wversions = None       # parameter versions the packed contents were built from

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

    _zeros = None        # a Model fills the exactly-zero bias gradients of all its blocks with one launch per step (Model.forward)
    _defer_nbt = False   # a Model bumps the num_batches_tracked counters of all its blocks with one multi-tensor add

    def nbt_buffers(self):
        return [self._tensor(bn).num_batches_tracked for bn in bn_names(self.cfg)]

    def _packed(self, params):
        """The block's packed weight forms for the current math mode.  The set (buffers + form table) is rebuilt when a parameter
        moved; its CONTENTS are re-packed when a parameter changed (version counters; FlatOptimizer.step and load_state_dict bump
        them) -- by the enclosing Model for all blocks in one launch, or here for a block used on its own."""
        mode = ops.get_math_mode()
        where = (mode,) + tuple(p.data_ptr() for p in params)
        if self._wcache is None or self._wcache[0] != where:
            self._wcache = (where, pack_weights(dict(zip(param_names(self.cfg), params)), self.cfg))
            self._wversions = None
        W = self._wcache[1]
        versions = tuple(p._version for p in params)
        if self._wversions != versions or not W.fresh:
            W.refresh()
            self._wversions = versions
        return W

    def mark_packed_stale(self) -> None:
        """Force a re-pack at the next forward (what an optimizer step does through the version counters)."""
        self._wversions = None

    def drop_packed(self) -> None:
        """Forget the packed set (parameters rewritten behind the version counters, e.g. by a collective)."""
        self._wcache = None
        self._wversions = None

    def recording_pins(self) -> list:
        """GraphStep hook: the packed set a recording made now reads (kept alive by the recording)."""
        return [] if self._wcache is None else [self._wcache[1]]

    def forward(self, x: torch.Tensor, pool_groups: int = 0, out_half: bool = False) -> torch.Tensor:
        """``pool_groups`` > 0 (the model's last block, block.pool_epilogue_ok): returns the block's output averaged over the rows of
        every group of consecutive samples, (pool_groups, out_channels), without forming the output.
        ``out_half``: the caller hands the output to another block of this kind, which takes a bfloat16 tensor -- in math mode bf16 with
        paths.half_activations (a training step) the output then IS bfloat16 (and its gradient arrives as one); float32 otherwise."""
        names = param_names(self.cfg)
        params = [self._tensor(n) for n in names]
        W = self._packed(params)
        holder = {"pool_groups": pool_groups} if pool_groups else {}
        if out_half and self.training and torch.is_grad_enabled():
            holder["out_half"] = True
        if not self.training and not torch.is_grad_enabled():
            holder["inference"] = True                # no backward can follow: the block may take its inference kernels (block_forward)
        if self._zeros is not None:                   # this step's slice of the model's zero pool (Model.forward), used once
            holder["zeros"], self._zeros = self._zeros, None
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


def refresh_packed_weights(model: nn.Module) -> None:
    """Re-pack the weight forms of ALL blocks whose parameters changed since they were packed, in ONE launch (packing.PackPlan
    over every live form of those blocks; the plan is cached on the model until the set of live forms changes).  Blocks that
    have no forms yet (first forward in this math mode) build theirs lazily inside their own forward."""
    from ...packing import PackPlan
    mode = ops.get_math_mode()
    stale = []
    for blk in model.modules():
        if not isinstance(blk, SpatialTemporalConv) or blk._wcache is None:
            continue
        params = [blk._tensor(n) for n in param_names(blk.cfg)]
        if blk._wcache[0] != (mode,) + tuple(p.data_ptr() for p in params):
            continue                                    # moved parameters / other math mode: the block rebuilds its set itself
        versions = tuple(p._version for p in params)
        if blk._wversions != versions:
            stale.append((blk, versions))
    if not stale:
        return
    forms = [f for blk, _ in stale for f in blk._wcache[1].live_forms()]
    sig = (mode, tuple(id(f) for f in forms))
    cached = getattr(model, "_pack_plan", None)
    if cached is None or cached[0] != sig:
        cached = (sig, PackPlan(forms))
        model._pack_plan = cached
    cached[1].run()
    for blk, versions in stale:
        blk._wcache[1].fresh = True
        blk._wversions = versions


def prepare_recording(model: nn.Module) -> None:
    """``GraphStep`` hook of the AGCN models (session/procedures/step.py): build what the first forward after an optimizer update
    builds lazily -- the one-launch re-pack plan over all blocks (device tables + an H2D copy, not recordable) -- before the
    step is recorded."""
    for blk in model.modules():
        if isinstance(blk, SpatialTemporalConv):
            blk.mark_packed_stale()
    refresh_packed_weights(model)


# (width multiplier of start_feature_size, temporal stride) of the ten blocks (reference :152-163); the first has no shortcut
BLOCK_PLAN = ((1, 1), (1, 1), (1, 1), (1, 1), (2, 2), (2, 1), (2, 1), (4, 2), (4, 1), (4, 1))


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
        kw = dict(static_adjacency=static_adjacency, fused_spatial=fused_spatial)
        # all ten blocks are constructed (and draw their initial weights) before the list is cut to num_layers, as in the
        # reference (:152-164): a shorter model built from the same seed then starts from the same parameters, `fc` included
        blocks, cin = [], num_channels
        for index, (mult, stride) in enumerate(BLOCK_PLAN):
            blocks.append(self._BLOCK(cin, start_feature_size * mult, adj, stride=stride, residual=index > 0, **kw))
            cin = blocks[-1].out_channels
        blocks = blocks[:min(len(blocks), num_layers)]
        cin = blocks[-1].out_channels
        # reference :166-172: with dropout, an in-place Dropout sits between consecutive blocks and takes an l<i> slot of its own
        self.layers = []
        for blk in blocks:
            if self.layers and dropout > 0:
                self.layers.append(nn.Dropout(dropout, inplace=True))
            self.layers.append(blk)
        for slot, layer in enumerate(self.layers):
            setattr(self, f"l{slot}", layer)
        self.fc = None if without_fc else nn.Linear(cin, num_classes)
        self.out_channels = cin if without_fc else num_classes
        if self.fc is not None:
            with torch.no_grad():
                self.fc.weight.normal_(0, math.sqrt(2. / num_classes))
        bn_init(self.data_bn, 1)

    def _blocks_input(self, x: torch.Tensor) -> torch.Tensor:
        """(N, M, T, V, C) -> data_bn over the (m, v, c) channels with statistics across (n, t)
        -> channels-last (N*M, T, V, C padded to 4): one statistics pass + one apply pass of libfgcn (block.data_bn); the
        reference's permute / view / BatchNorm1d / view / permute (:186-188) is the same function of x."""
        return data_bn(x, self.data_bn)

    def prepare_recording(self) -> None:
        prepare_recording(self)

    def recording_pins(self) -> list:
        """GraphStep hook: the one-launch re-pack plan (device tables) a recording made now replays."""
        plan = getattr(self, "_pack_plan", None)
        return [] if plan is None else [plan[1]]

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
        refresh_packed_weights(self)
        if self.training and torch.is_grad_enabled():
            # the exactly-zero bias gradients of all blocks (block._BiasGrads): one zero-filled allocation and one fill launch per step
            blocks = [m for m in self.layers if isinstance(m, SpatialTemporalConv)]
            sizes = [zero_bias_floats(b.cfg) for b in blocks]
            zeros = torch.zeros(sum(sizes), device=h.device, dtype=torch.float32)
            off = 0
            for b, n in zip(blocks, sizes):
                b._zeros = zeros[off:off + n]
                off += n
        for i, layer in enumerate(self.layers[:-1]):
            # (a block that feeds another block may hand over a bfloat16 tensor: math mode bf16, paths.half_activations)
            chain = isinstance(layer, SpatialTemporalConv) and isinstance(self.layers[i + 1], SpatialTemporalConv)
            h = layer(h, out_half=True) if chain else layer(h)
        # (N*M, T', V, C') -> mean over (T', V) then over persons = one mean over persons, frames and joints (equal-sized groups)
        last = self.layers[-1]
        if isinstance(last, SpatialTemporalConv) and pool_epilogue_ok(last.cfg, h.shape[0], h.shape[1], h.shape[2], N):
            h = last(h, pool_groups=N)                       # the block's epilogue pass sums per clip instead of writing its output
        else:
            h = last(h)
            h = GroupMeanFunction.apply(h.view(N, -1, h.size(-1)))
        if self.fc is not None:    # nn.Linear is the parameter container; the arithmetic is the row GEMM
            h = LinearFunction.apply(h.contiguous(), self.fc.weight, self.fc.bias)
        return h
