"""IMU-only graph model (reference torch_src/models/mmargcn/imu_feature_models.py:11-102): the IMU sequence
``(sequence_length, num_signals)`` becomes a graph with one node per value (or per sensor) and time step, connected inside a
time step and to the previous ``temporal_back_connections`` steps, and is classified by ``GCN``."""
import torch
import torch.nn as nn

from ...util.graph import Graph
from ...util.partition_strategy import GraphPartitionStrategy
from .gcn import GCN


def build_imu_graph(data_shape: tuple, num_signals: int = 0, temporal_back_connections: int = 1,
                    inter_signal_back_connections=False) -> Graph:
    """Vertices in memory order T0S0, T0S1, .., T0SM, T1S0, ..: all pairs inside a time step (both directions) and, per
    signal (or across signals), edges from the previous ``temporal_back_connections`` steps (reference :11-38)."""
    sequence_length, num_signals_0 = data_shape
    assert num_signals == 0 or (num_signals_0 % num_signals) == 0
    if num_signals == 0:
        num_signals = num_signals_0
    num_vertices = sequence_length * num_signals
    edges = []
    for i in range(0, num_vertices, num_signals):
        for j in range(num_signals):
            for k in range(j + 1, num_signals):
                edges.append((i + j, i + k))
                edges.append((i + k, i + j))
        for j in range(min(i // num_signals, temporal_back_connections)):
            for k in range(num_signals):
                for m in range(num_signals):
                    if k == m or inter_signal_back_connections:
                        edges.append((i - num_signals * (j + 1) + k, i + m))
    return Graph(edges, num_vertices)


def build_imu_graph_adjacency(data_shape: tuple, num_signals: int = 0, gc_model: str = "stgcn", sparse=False,
                              normalization="row", temporal_back_connections: int = 1,
                              inter_signal_back_connections: bool = False, build_graph_fn=build_imu_graph):
    graph = build_graph_fn(data_shape, num_signals, temporal_back_connections, inter_signal_back_connections)
    if gc_model == "agcn":
        return GraphPartitionStrategy().get_adjacency_matrix_array(graph)
    adj = graph.get_normalized_adjacency_matrix(normalization, True)     # (sparse=True: the same matrix, kept dense here)
    return torch.from_numpy(adj).to(torch.float32)


class ImuGCN(nn.Module):
    def __init__(self, data_shape, num_classes: int, **kwargs):
        super().__init__()
        data_shape = data_shape["inertial"]
        dropout = kwargs.get("dropout", 0.)
        sparse = kwargs.get("sparse", False)
        num_layers = kwargs.get("num_layers", 10)
        inner_feature_dim = kwargs.get("inner_feature_dim", 64)
        include_additional_top_layer = kwargs.get("include_additional_top_layer", False)
        num_temporal_back_connections = kwargs.get("num_temporal_back_connections", 1)
        inter_signal_back_connections = kwargs.get("inter_signal_back_connections", False)
        adjacency_normalization = kwargs.get("adjacency_normalization", "column")
        gc_model = kwargs.get("gc_model", "agcn")
        self.graph_node_format = kwargs.get("graph_node_format", "node_per_value")
        if self.graph_node_format == "node_per_value":
            num_signals = data_shape[1]
            self.num_features = 1
        elif self.graph_node_format == "node_per_sensor":
            num_signals = kwargs["num_signals"]
            self.num_features = data_shape[1] // num_signals
        else:
            raise ValueError(f"Unknown graph_node_format {self.graph_node_format}")
        num_nodes = data_shape[0] * num_signals
        adj = build_imu_graph_adjacency(data_shape, num_signals, gc_model, sparse, adjacency_normalization,
                                        num_temporal_back_connections, inter_signal_back_connections)
        self.gcn = GCN(adj, (self.num_features, num_nodes), num_classes, dropout, sparse, gc_model, num_layers,
                       inner_feature_dim, include_additional_top_layer, without_fc=kwargs.get("without_fc", False))

    def forward(self, x):
        if self.graph_node_format == "node_per_value":
            x = x.flatten(start_dim=1).unsqueeze(1).contiguous()
        else:
            x = x.view(x.shape[0], -1, self.num_features).permute(0, 2, 1).contiguous()
        return self.gcn(x)
