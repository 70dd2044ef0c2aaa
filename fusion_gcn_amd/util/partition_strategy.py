"""Partition strategy: skeleton graph -> (K, V, V) adjacency stack.

Mirrors ``GraphPartitionStrategy`` of the reference (util/partition_strategy.py:6-51).
Only the "spatial" strategy (K = 3) is functional in the reference ("distance" raises
NotImplementedError at :28, "uniform" raises TypeError at graph.py:114 via :51 — SURVEY.md
Appendix C item 2); the same error behaviour is kept here.
"""
from __future__ import annotations

import numpy as np

from .graph import Graph


class GraphPartitionStrategy:
    def __init__(self, strategy: str = "spatial"):
        assert strategy in ("uniform", "distance", "spatial")
        self.strategy = strategy

    def get_adjacency_matrix_array(self, graph: Graph, normalization: str = "column") -> np.ndarray:
        """(K, V, V) float64.  Spatial strategy, edges oriented (child, parent) towards the centre:
        A[0] = I (self), A[1] = column-normalised reversed directed graph (centripetal: every joint
        receives from its parent), A[2] = column-normalised directed graph (centrifugal: every joint
        receives the mean of its children)."""
        if self.strategy == "distance":
            raise NotImplementedError("Distance strategy not implemented since 'spatial' seems to yield best results.")
        if self.strategy == "spatial":
            directed = graph.as_directed()
            stack = np.empty((3, graph.num_vertices, graph.num_vertices))
            stack[0] = np.eye(graph.num_vertices)
            stack[1] = directed.with_reversed_edges().get_normalized_adjacency_matrix(normalization)
            stack[2] = directed.get_normalized_adjacency_matrix(normalization)
            return stack
        # "uniform": the reference passes normalization=True here and fails with TypeError
        # ('can only concatenate str (not "bool") to str'); keep that observable behaviour.
        raise TypeError('can only concatenate str (not "bool") to str')
