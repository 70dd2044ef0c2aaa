"""TEST INFRASTRUCTURE -- CPU restatement of the reference's MS-G3D model in stock torch ops.

NOT product code: only tests/ import this.  Parity status: PINNED against golden vectors generated from the imported reference
(oracle/gen_golden_msg3d.py -> tests/golden/msg3d.npz, checked by tests/test_msg3d.py).

What is restated is the reference's composition of torch ops (the arithmetic itself lives in its dependency ``torch``, pinned
1.6.0 in requirements.txt:13; 2.10.0 here), function by function:

  k_adjacency / normalize_adjacency   <- util/graph.py:176-184, models/msg3d/ms_gcn.py:17-21
  multi_scale_adjacency               <- MultiScale_GraphConv.__init__ (ms_gcn.py:37-45) / SpatialTemporal_MS_GCN (ms_gtcn.py:68-79)
  spatial_temporal_graph              <- SpatialTemporal_MS_GCN.build_spatial_temporal_graph (ms_gtcn.py:101-109)
  mlp                                 <- MLP.forward (mlp.py:14-30): Conv2d 1x1 + BatchNorm2d + activation
  ms_gcn                              <- MultiScale_GraphConv.forward (ms_gcn.py:53-64)
  temporal_conv / ms_tcn              <- TemporalConv (ms_tcn.py:15-34), MultiScale_TemporalConv.forward (ms_tcn.py:96-109)
  unfold_windows                      <- UnfoldTemporalWindows.forward (ms_gtcn.py:37-45)
  st_ms_gcn / ms_g3d / multi_window   <- SpatialTemporal_MS_GCN.forward (:111-126), MS_G3D.forward (msg3d.py:60-73),
                                         MultiWindow_MS_G3D.forward (:104-110)
  model_forward                       <- Model.forward (msg3d.py:155-182)

State is the flat ``dict[str, Tensor]`` of the reference's own state-dict keys; the constant adjacency stacks (plain attributes
in the reference, not in the state dict) are rebuilt from the binary adjacency.  Tensors are (N, C, T, V) like the reference.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
DILATIONS = (1, 2, 3, 4)
WINDOWS = ((3, 1), (5, 1))          # (window size, dilation) of the two G3D pathways


class Stats:
    def __init__(self):
        self.updates: Dict[str, Tensor] = {}


def batch_norm(x: Tensor, sd, prefix: str, train: bool, stats: Optional[Stats]) -> Tensor:
    gamma, beta = sd[f"{prefix}.weight"], sd[f"{prefix}.bias"]
    dims = [d for d in range(x.dim()) if d != 1]
    view = [1, -1] + [1] * (x.dim() - 2)
    if train:
        mean, var = x.mean(dim=dims), x.var(dim=dims, unbiased=False)
        if stats is not None:
            n = x.numel() // x.shape[1]
            rm, rv = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
            with torch.no_grad():
                stats.updates[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach().to(rm.dtype)
                stats.updates[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * (var.detach().to(rv.dtype) * (n / max(n - 1, 1)))
    else:
        mean, var = sd[f"{prefix}.running_mean"].to(x.dtype), sd[f"{prefix}.running_var"].to(x.dtype)
    return (x - mean.view(view)) * (torch.rsqrt(var + BN_EPS) * gamma).view(view) + beta.view(view)


# ---- graph constants -------------------------------------------------------------------------------------------------------
def k_adjacency(adj: np.ndarray, k: int) -> np.ndarray:
    """Nodes exactly k hops apart, plus the self loops (get_k_adjacency(..., with_self=True))."""
    eye = np.eye(len(adj), dtype=adj.dtype)
    if k == 0:
        return eye
    reach = lambda p: np.minimum(np.linalg.matrix_power(adj + eye, p), 1)      # noqa: E731
    return reach(k) - reach(k - 1) + eye


def normalize_adjacency(a: np.ndarray) -> np.ndarray:
    d = np.power(a.sum(-1), -0.5)
    return (np.eye(len(d)) * d @ a @ (np.eye(len(d)) * d)).astype(np.float32)


def multi_scale_adjacency(adj: np.ndarray, num_scales: int) -> np.ndarray:
    """(num_scales * V, V): the normalised exact-k-hop adjacencies stacked along the rows (disentangled aggregation)."""
    return np.concatenate([normalize_adjacency(k_adjacency(adj, k)) for k in range(num_scales)])


def spatial_temporal_graph(a_binary: np.ndarray, window: int) -> np.ndarray:
    return np.tile(a_binary + np.eye(len(a_binary), dtype=a_binary.dtype), (window, window)).copy()


# ---- layers ------------------------------------------------------------------------------------------------------------------
def mlp(x: Tensor, sd, p: str, train: bool, stats, relu: bool) -> Tensor:
    y = F.conv2d(x, sd[f"{p}.layers.0.weight"], sd[f"{p}.layers.0.bias"])
    y = batch_norm(y, sd, f"{p}.layers.1", train, stats)
    return torch.relu(y) if relu else y


def aggregate(x: Tensor, a: Tensor, num_scales: int) -> Tensor:
    """einsum('vu,nctu->nctv') with the stacked (S*V, V) matrix, scales moved into the channel axis: (N, S*C, T, V)."""
    n, c, t, v = x.shape
    agg = torch.einsum("vu,nctu->nctv", a, x).view(n, c, t, num_scales, v)
    return agg.permute(0, 3, 1, 2, 4).reshape(n, num_scales * c, t, v)


def ms_gcn(x: Tensor, sd, p: str, a_powers: Tensor, num_scales: int, train: bool, stats) -> Tensor:
    a = a_powers.to(x.dtype) + sd[f"{p}.A_res"].to(x.dtype)
    return mlp(aggregate(x, a, num_scales), sd, f"{p}.mlp", train, stats, relu=True)


def temporal_conv(x: Tensor, sd, p: str, stride: int, dilation: int, train: bool, stats) -> Tensor:
    w = sd[f"{p}.conv.weight"]
    k = w.shape[2]
    pad = (k + (k - 1) * (dilation - 1) - 1) // 2
    y = F.conv2d(x, w, sd[f"{p}.conv.bias"], stride=(stride, 1), padding=(pad, 0), dilation=(dilation, 1))
    return batch_norm(y, sd, f"{p}.bn", train, stats)


def ms_tcn(x: Tensor, sd, p: str, stride: int, train: bool, stats, final_relu: bool = True) -> Tensor:
    outs = []
    for i, dil in enumerate(DILATIONS):
        b = f"{p}.branches.{i}"
        h = torch.relu(batch_norm(F.conv2d(x, sd[f"{b}.0.weight"], sd[f"{b}.0.bias"]), sd, f"{b}.1", train, stats))
        outs.append(temporal_conv(h, sd, f"{b}.3", stride, dil, train, stats))
    b = f"{p}.branches.4"
    h = torch.relu(batch_norm(F.conv2d(x, sd[f"{b}.0.weight"], sd[f"{b}.0.bias"]), sd, f"{b}.1", train, stats))
    h = F.max_pool2d(h, kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0))
    outs.append(batch_norm(h, sd, f"{b}.4", train, stats))
    b = f"{p}.branches.5"
    outs.append(batch_norm(F.conv2d(x, sd[f"{b}.0.weight"], sd[f"{b}.0.bias"], stride=(stride, 1)), sd, f"{b}.1", train, stats))
    out = torch.cat(outs, dim=1)
    if f"{p}.residual.conv.weight" in sd:
        out = out + temporal_conv(x, sd, f"{p}.residual", stride, 1, train, stats)
    elif out.shape == x.shape:
        out = out + x
    else:                      # residual=False never occurs in the model
        raise AssertionError("ms_tcn: no residual branch for a shape-changing block")
    return torch.relu(out) if final_relu else out


def unfold_windows(x: Tensor, window: int, stride: int, dilation: int) -> Tensor:
    n, c, t, v = x.shape
    pad = (window + (window - 1) * (dilation - 1) - 1) // 2
    u = F.unfold(x, kernel_size=(window, 1), dilation=(dilation, 1), stride=(stride, 1), padding=(pad, 0))
    u = u.view(n, c, window, -1, v).permute(0, 1, 3, 2, 4).contiguous()
    return u.view(n, c, -1, window * v)


def ms_g3d(x: Tensor, sd, p: str, a_scales: Tensor, num_scales: int, window: int, stride: int, dilation: int, train: bool, stats) -> Tensor:
    n, _, _, v = x.shape
    xw = unfold_windows(x, window, stride, dilation)                        # (N, C, T', window * V)
    q = f"{p}.gcn3d.1"
    a = a_scales.to(x.dtype) + sd[f"{q}.A_res"].to(x.dtype)
    h = torch.relu(mlp(aggregate(xw, a, num_scales), sd, f"{q}.mlp", train, stats, relu=False))
    c_embed = h.shape[1]
    h = h.view(n, c_embed, -1, window, v)
    y = F.conv3d(h, sd[f"{p}.out_conv.weight"], sd[f"{p}.out_conv.bias"]).squeeze(3)
    return batch_norm(y, sd, f"{p}.out_bn", train, stats)


def model_forward(x: Tensor, sd, a_binary: np.ndarray, train: bool = True, stats: Optional[Stats] = None, num_gcn_scales: int = 13,
                  num_g3d_scales: int = 6) -> Tensor:
    """x: (N, M, T, V, C) -> logits."""
    n, m, t, v, c = x.shape
    a_gcn = torch.from_numpy(multi_scale_adjacency(a_binary.astype(np.float64), num_gcn_scales))
    a_g3d = {w: torch.from_numpy(multi_scale_adjacency(spatial_temporal_graph(a_binary.astype(np.float64), w), num_g3d_scales))
             for w, _ in WINDOWS}
    h = x.permute(0, 1, 3, 4, 2).reshape(n, m * v * c, t)
    h = batch_norm(h, sd, "data_bn", train, stats)
    h = h.view(n * m, v, c, t).permute(0, 2, 3, 1).contiguous()
    for i, stride in ((1, 1), (2, 2), (3, 2)):
        s = ms_gcn(h, sd, f"sgcn{i}.0", a_gcn, num_gcn_scales, train, stats)
        s = ms_tcn(s, sd, f"sgcn{i}.1", stride, train, stats)
        s = ms_tcn(s, sd, f"sgcn{i}.2", 1, train, stats, final_relu=False)
        g = 0
        for j, (w, dil) in enumerate(WINDOWS):
            g = g + ms_g3d(h, sd, f"gcn3d{i}.gcn3d.{j}", a_g3d[w], num_g3d_scales, w, stride, dil, train, stats)
        h = torch.relu(s + g)
        h = ms_tcn(h, sd, f"tcn{i}", 1, train, stats)
    feat = h.view(n, m, h.shape[1], -1).mean(3).mean(1)
    return F.linear(feat, sd["fc.weight"], sd["fc.bias"])


def loss_and_grads(x: Tensor, labels: Tensor, sd, a_binary: np.ndarray, train: bool = True):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    full = dict(sd)
    full.update(params)
    stats = Stats()
    logits = model_forward(x, full, a_binary, train=train, stats=stats)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(params.keys(), grads)), stats
