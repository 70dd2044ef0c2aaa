"""TEST INFRASTRUCTURE — CPU restatement of the reference's AGCN / ST-GCN hot path in stock torch ops.

NOT product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Parity status: PINNED against golden vectors generated from the imported reference
(oracle/gen_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).

The reference's arithmetic lives in its third-party dependency ``torch`` (pinned torch==1.6.0 in the
reference's requirements.txt:13; this image has 2.10.0).  What is restated here is the reference's own
composition of those ops, function by function (formulas: SURVEY.md Appendix A):

  spatial_graph_conv   <- SpatialGraphConv.forward      torch_src/models/mmargcn/agcn.py:96-115
  temporal_conv        <- TemporalConv.forward          torch_src/models/mmargcn/agcn.py:49-51
  st_block             <- SpatialTemporalConv.forward   torch_src/models/mmargcn/agcn.py:134-136
  model_forward        <- Model.forward                 torch_src/models/mmargcn/agcn.py:183-200
  block_plan           <- Model.__init__ layer table    torch_src/models/mmargcn/agcn.py:152-164
  new_state_dict       <- module construction / names   torch_src/models/mmargcn/agcn.py:54-94,118-181

State is a flat ``dict[str, Tensor]`` using the reference's own state-dict keys for
``models.mmargcn.agcn.Model`` (``l0.gcn1.conv_a.0.weight`` ...); ``agcn_key`` maps them to the
``models.agcn.agcn.Model`` spelling (``l1.gcn1.PA`` ...).  Tensors are NCHW = (B, C, T, V) like the reference.
Works in float32 or float64 (dtype follows the inputs).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
NUM_SUBSETS = 3


# ------------------------------------------------------------------------------------------------
# structure
# ------------------------------------------------------------------------------------------------
def block_plan(num_channels: int, num_layers: int = 10, start: int = 64) -> List[dict]:
    """(cin, cout, stride, residual) of the 10 blocks (reference agcn.py:152-164)."""
    widths = [start] * 4 + [2 * start] * 3 + [4 * start] * 3
    plan, cin = [], num_channels
    for i, cout in enumerate(widths):
        stride = 2 if i in (4, 7) else 1
        plan.append(dict(cin=cin, cout=cout, stride=stride, residual=(i != 0)))
        cin = cout
    return plan[:min(len(plan), num_layers)]


def _bn_entries(prefix: str, c: int, gamma: float) -> Dict[str, Tensor]:
    return {f"{prefix}.weight": torch.full((c,), gamma), f"{prefix}.bias": torch.zeros(c),
            f"{prefix}.running_mean": torch.zeros(c), f"{prefix}.running_var": torch.ones(c),
            f"{prefix}.num_batches_tracked": torch.zeros((), dtype=torch.long)}


def new_state_dict(data_shape, num_classes: int, adjacency: np.ndarray, num_layers: int = 10,
                   start: int = 64, without_fc: bool = False, dtype=torch.float32) -> Dict[str, Tensor]:
    """All parameters/buffers with the reference's key names, shapes and *constant* init values
    (BN gamma 1 / 1e-6, adj_b 1e-6, biases 0 — agcn.py:62-63,86-94); random-init weights are left at 0
    because every test fills them through oracle.filler."""
    m, _t, v, c = data_shape
    sd: Dict[str, Tensor] = {}
    sd.update(_bn_entries("data_bn", m * v * c, 1.0))
    adj = torch.from_numpy(np.asarray(adjacency, dtype=np.float32))
    for i, b in enumerate(block_plan(c, num_layers, start)):
        cin, cout, p = b["cin"], b["cout"], f"l{i}"
        ic = cout // 4
        sd[f"{p}.gcn1.adj_b"] = torch.full_like(adj, 1e-6)
        sd[f"{p}.gcn1.adj_a"] = adj.clone()
        for name, oc in (("conv_a", ic), ("conv_b", ic), ("conv_d", cout)):
            for k in range(NUM_SUBSETS):
                sd[f"{p}.gcn1.{name}.{k}.weight"] = torch.zeros(oc, cin, 1, 1)
                sd[f"{p}.gcn1.{name}.{k}.bias"] = torch.zeros(oc)
        if cin != cout:
            sd[f"{p}.gcn1.down.0.weight"] = torch.zeros(cout, cin, 1, 1)
            sd[f"{p}.gcn1.down.0.bias"] = torch.zeros(cout)
            sd.update(_bn_entries(f"{p}.gcn1.down.1", cout, 1.0))
        sd.update(_bn_entries(f"{p}.gcn1.bn", cout, 1e-6))
        sd[f"{p}.tcn1.conv.weight"] = torch.zeros(cout, cout, 9, 1)
        sd[f"{p}.tcn1.conv.bias"] = torch.zeros(cout)
        sd.update(_bn_entries(f"{p}.tcn1.bn", cout, 1.0))
        if b["residual"] and not (cin == cout and b["stride"] == 1):
            sd[f"{p}.residual.conv.weight"] = torch.zeros(cout, cin, 1, 1)
            sd[f"{p}.residual.conv.bias"] = torch.zeros(cout)
            sd.update(_bn_entries(f"{p}.residual.bn", cout, 1.0))
    if not without_fc:
        last = block_plan(c, num_layers, start)[-1]["cout"]
        sd["fc.weight"] = torch.zeros(num_classes, last)
        sd["fc.bias"] = torch.zeros(num_classes)
    # reference registration order interleaves differently; order is irrelevant for a dict
    return {k: (t.to(dtype) if t.is_floating_point() else t) for k, t in sd.items()}


def agcn_key(key: str) -> str:
    """mmargcn-style key -> models/agcn/agcn.py key (l0..l9 -> l1..l10, adj_b -> PA; no adj_a entry)."""
    head, _, rest = key.partition(".")
    if head.startswith("l") and head[1:].isdigit():
        head = f"l{int(head[1:]) + 1}"
    out = f"{head}.{rest}" if rest else head
    return out.replace(".adj_b", ".PA")


# ------------------------------------------------------------------------------------------------
# ops
# ------------------------------------------------------------------------------------------------
class Stats:
    """Collects the BatchNorm running-stat updates of one training forward (momentum 0.1, unbiased var)."""

    def __init__(self):
        self.updates: Dict[str, Tensor] = {}


def batch_norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, train: bool, stats: Optional[Stats]) -> Tensor:
    """nn.BatchNorm{1,2}d semantics on dim 1; train: batch statistics (biased var) + running update."""
    gamma, beta = sd[f"{prefix}.weight"], sd[f"{prefix}.bias"]
    dims = [d for d in range(x.dim()) if d != 1]
    view = [1, -1] + [1] * (x.dim() - 2)
    if train:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if stats is not None:
            n = x.numel() // x.shape[1]
            rm, rv = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
            with torch.no_grad():
                stats.updates[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach().to(rm.dtype)
                stats.updates[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * (
                    var.detach().to(rv.dtype) * (n / max(n - 1, 1)))
                stats.updates[f"{prefix}.num_batches_tracked"] = sd[f"{prefix}.num_batches_tracked"] + 1
    else:
        mean, var = sd[f"{prefix}.running_mean"].to(x.dtype), sd[f"{prefix}.running_var"].to(x.dtype)
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(view)) * (inv * gamma).view(view) + beta.view(view)


def effective_adjacency(x: Tensor, sd, p: str, k: int) -> Tuple[Tensor, Tensor]:
    """C_k = column-softmax of the embedded-Gaussian joint affinity, and A_k + B_k + C_k  (agcn.py:100-108)."""
    b, _c, t, v = x.shape
    theta = F.conv2d(x, sd[f"{p}.conv_a.{k}.weight"], sd[f"{p}.conv_a.{k}.bias"])   # (B, ic, T, V)
    phi = F.conv2d(x, sd[f"{p}.conv_b.{k}.weight"], sd[f"{p}.conv_b.{k}.bias"])
    ic = theta.shape[1]
    left = theta.permute(0, 3, 1, 2).reshape(b, v, ic * t)
    right = phi.reshape(b, ic * t, v)
    score = torch.matmul(left, right) / (ic * t)
    c_k = torch.softmax(score, dim=-2)
    return c_k, c_k + (sd[f"{p}.adj_a"][k] + sd[f"{p}.adj_b"][k]).to(x.dtype)


def spatial_graph_conv(x: Tensor, sd, p: str, train: bool, stats: Optional[Stats] = None,
                       static_adjacency: bool = False):
    """G = ReLU(BN(sum_k Wd_k (x A^_k) + bd_k) + down(x)); returns (G, [C_0, C_1, C_2]).
    ``static_adjacency`` drops the data-dependent C_k term (the ST-GCN special case)."""
    b, c, t, v = x.shape
    y, adj_c = None, []
    for k in range(NUM_SUBSETS):
        if static_adjacency:
            a_hat = (sd[f"{p}.adj_a"][k] + sd[f"{p}.adj_b"][k]).to(x.dtype).expand(b, v, v)
            adj_c.append(None)
        else:
            c_k, a_hat = effective_adjacency(x, sd, p, k)
            adj_c.append(c_k)
        agg = torch.matmul(x.reshape(b, c * t, v), a_hat).reshape(b, c, t, v)
        z = F.conv2d(agg, sd[f"{p}.conv_d.{k}.weight"], sd[f"{p}.conv_d.{k}.bias"])
        y = z if y is None else z + y
    y = batch_norm(y, sd, f"{p}.bn", train, stats)
    if f"{p}.down.0.weight" in sd:
        d = F.conv2d(x, sd[f"{p}.down.0.weight"], sd[f"{p}.down.0.bias"])
        y = y + batch_norm(d, sd, f"{p}.down.1", train, stats)
    else:
        y = y + x
    return torch.relu(y), adj_c


def temporal_conv(x: Tensor, sd, p: str, stride: int, train: bool, stats: Optional[Stats] = None) -> Tensor:
    """BN(Conv2d((kt,1), pad ((kt-1)//2, 0), stride (s,1)))  — no activation (agcn.py:41-51)."""
    w = sd[f"{p}.conv.weight"]
    pad = (w.shape[2] - 1) // 2
    u = F.conv2d(x, w, sd[f"{p}.conv.bias"], stride=(stride, 1), padding=(pad, 0))
    return batch_norm(u, sd, f"{p}.bn", train, stats)


def st_block(x: Tensor, sd, p: str, stride: int, residual: bool, train: bool, stats: Optional[Stats] = None,
             static_adjacency: bool = False, capture: Optional[dict] = None):
    """``capture``: receives the two ReLU outputs of the block, ``{p}.g`` and ``{p}.o`` (detached) -- the ReLU decisions
    the model-level gradient-parity test compares and injects (oracle/relu_masks.py)."""
    g, adj_c = spatial_graph_conv(x, sd, f"{p}.gcn1", train, stats, static_adjacency)
    z = temporal_conv(g, sd, f"{p}.tcn1", stride, train, stats)
    if not residual:
        res = 0
    elif f"{p}.residual.conv.weight" in sd:
        res = temporal_conv(x, sd, f"{p}.residual", stride, train, stats)
    else:
        res = x
    o = torch.relu(z + res)
    if capture is not None:
        capture[f"{p}.g"], capture[f"{p}.o"] = g.detach(), o.detach()
    return o, adj_c


def model_forward(x: Tensor, sd, train: bool = True, stats: Optional[Stats] = None, num_layers: int = 10,
                  start: int = 64, static_adjacency: bool = False, return_blocks: bool = False,
                  capture: Optional[dict] = None):
    """x: (N, M, T, V, C) -> logits (N, classes) (or pooled features if the dict has no ``fc``)."""
    n, m, t, v, c = x.shape
    h = x.permute(0, 1, 3, 4, 2).reshape(n, m * v * c, t)
    h = batch_norm(h, sd, "data_bn", train, stats)
    h = h.reshape(n, m, v, c, t).permute(0, 1, 3, 4, 2).reshape(n * m, c, t, v)
    blocks = []
    for i, b in enumerate(block_plan(c, num_layers, start)):
        h, _ = st_block(h, sd, f"l{i}", b["stride"], b["residual"], train, stats, static_adjacency, capture)
        if return_blocks:
            blocks.append(h)
    feat = h.reshape(n, m, h.shape[1], -1).mean(3).mean(1)
    out = F.linear(feat, sd["fc.weight"], sd["fc.bias"]) if "fc.weight" in sd else feat
    return (out, blocks) if return_blocks else out


def loss_and_grads(x: Tensor, labels: Tensor, sd, train: bool = True, **kw):
    """CrossEntropy (mean) forward + backward like DefaultStep (reference session/procedures/step.py:38-46).
    Returns (logits, loss, {param key: grad}, Stats)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "adj_a"))}
    full = dict(sd)
    full.update(params)
    stats = Stats()
    logits = model_forward(x, full, train=train, stats=stats, **kw)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(params.keys(), grads)), stats


def conv_param_init_std(kind: str, out_c: int, in_c: int, kt: int = 1, branches: int = 3) -> float:
    """Std of the reference's random inits (agcn.py:18-29): kaiming fan_out for convs, branch init for conv_d."""
    if kind == "branch":
        return math.sqrt(2.0 / (out_c * in_c * kt * branches))
    return math.sqrt(2.0 / (out_c * kt))
