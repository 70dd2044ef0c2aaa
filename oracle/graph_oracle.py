"""TEST INFRASTRUCTURE — loop-level restatement of the reference's adjacency construction.

Follows, step by step and independently of ``fusion_gcn_amd.util``:
  * Graph.get_adjacency_matrix            reference util/graph.py:74-79
  * Graph.get_normalized_adjacency_matrix reference util/graph.py:116-124 (+ :97-114)
  * GraphPartitionStrategy (spatial)      reference util/partition_strategy.py:36-46
  * get_skeleton_imu_fusion_graph         reference torch_src/models/mmargcn/fusion.py:65-89
Plain Python loops over tiny matrices (V <= 27).
"""
from __future__ import annotations

import numpy as np


def binary_adjacency(edges, num_vertices: int, directed: bool) -> np.ndarray:
    a = np.zeros((num_vertices, num_vertices))
    for i, j in edges:
        a[int(i), int(j)] = 1.0
        if not directed:
            a[int(j), int(i)] = 1.0
    return a


def column_normalised(adj: np.ndarray) -> np.ndarray:
    v = adj.shape[0]
    out = np.zeros_like(adj)
    for col in range(v):
        deg = adj[:, col].sum()
        if deg > 0:
            for row in range(v):
                out[row, col] = adj[row, col] / deg
    return out


def spatial_partition_stack(edges, num_vertices=None) -> np.ndarray:
    """(3, V, V): identity; column-normalised reversed directed graph; column-normalised directed graph."""
    edges = [(int(a), int(b)) for a, b in edges]
    if num_vertices is None:
        num_vertices = max(max(e) for e in edges) + 1
    fwd = binary_adjacency(edges, num_vertices, directed=True)
    rev = binary_adjacency([(b, a) for a, b in edges], num_vertices, directed=True)
    return np.stack([np.eye(num_vertices), column_normalised(rev), column_normalised(fwd)])


def imu_fusion_edges(edges, num_vertices: int, mode: str, num_imu_joints: int, center_joint=None,
                     right_wrist_joint=None, right_hip_joint=None, interconnect: bool = False):
    """Edge list of the skeleton graph extended by IMU joints (vertices V .. V+n-1)."""
    out = [(int(a), int(b)) for a, b in edges]
    for i in range(num_imu_joints):
        if mode == "append_center":
            out.append((num_vertices + i, center_joint))
        elif mode == "append_right":
            out.append((num_vertices + i, right_wrist_joint))
            out.append((num_vertices + i, right_hip_joint))
        else:
            raise ValueError("Unsupported imu_enhanced_mode: " + mode)
    if interconnect:
        for i in range(num_imu_joints):
            for j in range(i + 1, num_imu_joints):
                out.append((num_vertices + i, num_vertices + j))
    # the reference rebuilds the graph through np.unique(edges, axis=0): duplicates collapse
    return sorted(set(out))
