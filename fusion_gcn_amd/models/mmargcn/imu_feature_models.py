"""IMU-only graph model (reference torch_src/models/mmargcn/imu_feature_models.py:11-102): the IMU sequence
``(sequence_length, num_signals)`` becomes a graph with one node per value (or per sensor) and time step, connected inside a
time step and to the previous ``temporal_back_connections`` steps, and is classified by ``GCN``."""
import numpy as np
import torch
import torch.nn as nn

from ...util.graph import Graph
from ...util.partition_strategy import GraphPartitionStrategy
from .gcn import GCN


def build_imu_graph(data_shape: tuple, num_signals: int = 0, temporal_back_connections: int = 1,
                    inter_signal_back_connections=False) -> Graph:
    """Vertex (t, s) = t * S + s.  Edges: every ordered pair of distinct signals inside a time step; from step t - d
    (d = 1 .. temporal_back_connections, t - d >= 0) to step t, signal to the same signal or -- ``inter_signal_back_connections`` --
    to every signal (reference :11-38, built there with nested loops; here as index grids)."""
    steps, total = data_shape
    S = num_signals or total
    if total % S:
        raise AssertionError("the signal count must divide the sequence's width")
    t = np.arange(steps)[:, None, None]
    a, b = np.arange(S)[None, :, None], np.arange(S)[None, None, :]
    inside = np.broadcast_to(a != b, (steps, S, S))
    edges = [np.stack([np.broadcast_to(t * S + a, inside.shape)[inside], np.broadcast_to(t * S + b, inside.shape)[inside]], 1)]
    link = np.broadcast_to((a == b) | bool(inter_signal_back_connections), (steps, S, S))
    for d in range(1, temporal_back_connections + 1):
        ok = link & (t - d >= 0)
        edges.append(np.stack([np.broadcast_to((t - d) * S + a, ok.shape)[ok], np.broadcast_to(t * S + b, ok.shape)[ok]], 1))
    return Graph(np.concatenate(edges).astype(np.int64), steps * S)


def build_imu_graph_adjacency(data_shape: tuple, num_signals: int = 0, gc_model: str = "stgcn", sparse=False,
                              normalization="row", temporal_back_connections: int = 1,
                              inter_signal_back_connections: bool = False, build_graph_fn=build_imu_graph):
    """``agcn``: the (3, V, V) partition-strategy stack; otherwise the self-connected, normalised (V, V) adjacency as a float32
    tensor (the reference's sparse variant holds the same matrix; kept dense here)."""
    graph = build_graph_fn(data_shape, num_signals, temporal_back_connections, inter_signal_back_connections)
    if gc_model == "agcn":
        return GraphPartitionStrategy().get_adjacency_matrix_array(graph)
    return torch.from_numpy(graph.get_normalized_adjacency_matrix(normalization, True)).to(torch.float32)


_IMU_GCN_DEFAULTS = dict(dropout=0., sparse=False, num_layers=10, inner_feature_dim=64, include_additional_top_layer=False,
                         num_temporal_back_connections=1, inter_signal_back_connections=False,
                         adjacency_normalization="column", gc_model="agcn", graph_node_format="node_per_value", without_fc=False)


class ImuGCN(nn.Module):
    def __init__(self, data_shape, num_classes: int, **kwargs):
        super().__init__()
        steps, width = data_shape["inertial"]
        o = {**_IMU_GCN_DEFAULTS, **{k: v for k, v in kwargs.items() if k in _IMU_GCN_DEFAULTS}}
        self.graph_node_format = o["graph_node_format"]
        if self.graph_node_format == "node_per_value":       # every scalar of the sequence is a node with one feature
            signals, self.num_features = width, 1
        elif self.graph_node_format == "node_per_sensor":    # a sensor's axes are the features of its node
            signals = kwargs["num_signals"]
            self.num_features = width // signals
        else:
            raise ValueError(f"Unknown graph_node_format {self.graph_node_format}")
        adj = build_imu_graph_adjacency((steps, width), signals, o["gc_model"], o["sparse"], o["adjacency_normalization"],
                                        o["num_temporal_back_connections"], o["inter_signal_back_connections"])
        self.gcn = GCN(adj, (self.num_features, steps * signals), num_classes, o["dropout"], o["sparse"], o["gc_model"],
                       o["num_layers"], o["inner_feature_dim"], o["include_additional_top_layer"], without_fc=o["without_fc"])

    def forward(self, x):
        n = x.shape[0]
        if self.num_features == 1:
            x = x.reshape(n, 1, -1)
        else:
            x = x.reshape(n, -1, self.num_features).transpose(1, 2)
        return self.gcn(x.contiguous())
