"""TEST INFRASTRUCTURE -- CPU restatement of the reference's IMU graph model with the static-adjacency 1-D graph convolution
(torch_src/models/mmargcn/imu_feature_models.py:11-102, gcn.py:18-83, graph_convolution.py:12-52), as plain functions over a
state dict with the reference's key names.  Stock torch CPU ops, any float dtype.  Pinned to tests/golden/imu_gcn.npz, which
oracle/gen_golden_imu.py writes by importing the reference itself.  Only tests may import this module."""
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def imu_graph_edges(data_shape: Tuple[int, int], num_signals: int = 0, temporal_back_connections: int = 1,
                    inter_signal_back_connections: bool = False) -> Tuple[List[Tuple[int, int]], int]:
    """build_imu_graph (imu_feature_models.py:11-38): -> (edge list, number of vertices)."""
    sequence_length, num_signals_0 = data_shape
    if num_signals == 0:
        num_signals = num_signals_0
    n = sequence_length * num_signals
    edges = []
    for i in range(0, n, num_signals):
        for j in range(num_signals):
            for k in range(j + 1, num_signals):
                edges += [(i + j, i + k), (i + k, i + j)]
        for j in range(min(i // num_signals, temporal_back_connections)):
            for k in range(num_signals):
                for m in range(num_signals):
                    if k == m or inter_signal_back_connections:
                        edges.append((i - num_signals * (j + 1) + k, i + m))
    return edges, n


def normalized_adjacency(edges, n: int, normalization: str = "row", self_connections: bool = True) -> np.ndarray:
    """Graph.get_normalized_adjacency_matrix of an undirected graph (util/graph.py:74-79,97-124), float64."""
    a = np.zeros((n, n))
    for i, j in edges:
        a[i, j] = 1.0
        a[j, i] = 1.0
    if self_connections:
        a += np.eye(n)
    d = a.sum(axis=0)
    d_inv = np.zeros_like(d)
    nz = d > 0
    d_inv[nz] = 1.0 / (np.sqrt(d[nz]) if normalization == "symmetric" else d[nz])
    dm = np.diag(d_inv)
    if normalization == "row":
        return dm @ a
    if normalization == "column":
        return a @ dm
    if normalization in ("row_column", "symmetric"):
        return dm @ a @ dm
    raise ValueError(normalization)


def _bn(x: Tensor, sd: Dict[str, Tensor], p: str, train: bool) -> Tensor:
    return F.batch_norm(x, None if train else sd[p + "running_mean"], None if train else sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], training=train, momentum=0.1, eps=1e-5)


def gcn_forward(x: Tensor, sd: Dict[str, Tensor], prefix: str = "gcn.", train: bool = True) -> Tensor:
    """GCN.forward (gcn.py:70-83) with STGCNGraphConvolution layers: x (B, F, V) -> logits (or pooled features)."""
    b, f, v = x.shape
    h = _bn(torch.flatten(x, start_dim=1), sd, prefix + "bn.", train).reshape(b, f, v)
    i = 1
    while f"{prefix}gc{i}.conv_a.0.weight" in sd:          # AGCNGraphConvolution layers (graph_convolution.py:91-113)
        p = f"{prefix}gc{i}."
        adj = (sd[p + "adj_a"] + sd[p + "adj_b"]).to(h.dtype)
        y = None
        for k in range(3):
            a1 = F.conv1d(h, sd[f"{p}conv_a.{k}.weight"], sd[f"{p}conv_a.{k}.bias"]).permute(0, 2, 1)
            a2 = F.conv1d(h, sd[f"{p}conv_b.{k}.weight"], sd[f"{p}conv_b.{k}.bias"])
            a1 = torch.softmax(torch.matmul(a1, a2) / a1.size(-1), dim=-2) + adj[k]
            z = F.conv1d(torch.matmul(h, a1), sd[f"{p}conv_d.{k}.weight"], sd[f"{p}conv_d.{k}.bias"])
            y = z + y if y is not None else z
        y = _bn(y, sd, p + "bn.", train)
        if p + "down.0.weight" in sd:
            y = y + _bn(F.conv1d(h, sd[p + "down.0.weight"], sd[p + "down.0.bias"]), sd, p + "down.1.", train)
        else:
            y = y + h
        h = torch.relu(y)
        i += 1
    while f"{prefix}gc{i}.conv.weight" in sd:
        p = f"{prefix}gc{i}."
        support = F.conv1d(h, sd[p + "conv.weight"], sd[p + "conv.bias"])
        out = torch.matmul(support, sd[p + "adj"].to(h.dtype).t())
        if p + "residual.0.weight" in sd:
            res = _bn(F.conv1d(h, sd[p + "residual.0.weight"], sd[p + "residual.0.bias"]), sd, p + "residual.1.", train)
        elif i > 1 and h.shape[1] == out.shape[1]:      # (the first layer is built with residual=False, gcn.py:42-44)
            res = h
        else:
            res = 0
        h = torch.relu(out + res)
        i += 1
    h = h.mean(-1)
    if prefix + "fc.weight" in sd:
        h = F.linear(h, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    return h


def imu_gcn_forward(x: Tensor, sd: Dict[str, Tensor], graph_node_format: str = "node_per_value", num_features: int = 1,
                    train: bool = True) -> Tensor:
    """ImuGCN.forward (imu_feature_models.py:91-99): x (B, T, signals)."""
    if graph_node_format == "node_per_value":
        x = x.flatten(start_dim=1).unsqueeze(1)
    else:
        x = x.reshape(x.shape[0], -1, num_features).permute(0, 2, 1)
    return gcn_forward(x.contiguous(), sd, "gcn.", train)


def loss_and_grads(x: Tensor, labels: Tensor, sd: Dict[str, Tensor], **kw):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var", ".adj", ".adj_a"))}
    full = dict(sd)
    full.update(params)
    logits = imu_gcn_forward(x, full, **kw)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(params.keys(), grads))
