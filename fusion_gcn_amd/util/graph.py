"""Skeleton graph -> adjacency matrices (host side, build time only).

Mirrors the interface of the reference's ``util/graph.py`` (``Graph``: reference
util/graph.py:7-173) for the part the AGCN / ST-GCN hot path uses: edge list in,
binary adjacency / degree / normalised adjacency out.  Pure numpy; none of the
reference's import-time dependencies (networkx, matplotlib, scipy) are needed.

Behavioural notes that matter for parity:
  * edges are de-duplicated and sorted row-wise like ``np.unique(edges, axis=0)``
    (reference util/graph.py:13);
  * ``with_new_edges`` returns an UNDIRECTED graph and re-derives ``num_vertices``
    from the edge list (reference util/graph.py:60-65);
  * zero-degree vertices get a reciprocal degree of 0 (the reference leaves them
    uninitialised, util/graph.py:98-101; they only ever multiply all-zero columns).
"""
from __future__ import annotations

import numpy as np

_NORMALIZATIONS = ("row", "column", "row_column", "symmetric")


class Graph:
    """Edge-list graph. ``edges[i] = (a, b)`` means an edge a -> b when directed."""

    def __init__(self, edges, num_vertices=None, is_directed: bool = False, center_joint: int = 0):
        e = np.asarray(edges)
        if e.ndim != 2 or e.shape[1] != 2:
            raise AssertionError("edges must have shape (E, 2)")
        if not np.issubdtype(e.dtype, np.integer):
            raise AssertionError("edges must be integers")
        if np.any(e < 0):
            raise AssertionError("edges must be non-negative")
        self.edges = np.unique(e, axis=0)
        needed = int(self.edges.max()) + 1
        if num_vertices is None:
            self.num_vertices = needed
        else:
            if num_vertices < needed:
                raise AssertionError("num_vertices smaller than largest vertex id + 1")
            self.num_vertices = int(num_vertices)
        self.is_directed = bool(is_directed)
        self.center_joint = center_joint

    # -- views -----------------------------------------------------------------------------
    def as_directed(self) -> "Graph":
        return self if self.is_directed else Graph(self.edges, self.num_vertices, True, self.center_joint)

    def as_undirected(self) -> "Graph":
        return Graph(self.edges, self.num_vertices, False, self.center_joint) if self.is_directed else self

    def with_reversed_edges(self) -> "Graph":
        return Graph(self.edges[:, ::-1].copy(), self.num_vertices, self.is_directed, self.center_joint)

    def with_new_edges(self, edges) -> "Graph":
        extra = np.asarray(edges)
        if extra.size == 0:
            extra = np.zeros((0, 2), dtype=self.edges.dtype)
        if extra.ndim != 2 or extra.shape[1] != 2 or not np.issubdtype(extra.dtype, np.integer) or np.any(extra < 0):
            raise AssertionError("edges must be non-negative integers of shape (E, 2)")
        # reference quirk: result is undirected and num_vertices is recomputed
        return Graph(np.vstack((self.edges, extra)), center_joint=self.center_joint)

    def with_removed_edges(self, edges) -> "Graph":
        drop = {tuple(int(x) for x in e) for e in np.asarray(edges)}
        keep = np.array([tuple(e) not in drop for e in self.edges.tolist()], dtype=bool)
        return Graph(self.edges[keep], center_joint=self.center_joint)

    def has_edge(self, edge) -> bool:
        a, b = edge
        return bool(np.any((self.edges[:, 0] == a) & (self.edges[:, 1] == b)))

    def has_edges(self, edges) -> np.ndarray:
        return np.array([self.has_edge(e) for e in np.asarray(edges)], dtype=bool)

    # -- matrices --------------------------------------------------------------------------
    def get_adjacency_matrix(self) -> np.ndarray:
        a = np.zeros((self.num_vertices, self.num_vertices), dtype=np.int64)
        a[self.edges[:, 0], self.edges[:, 1]] = 1
        if not self.is_directed:
            a[self.edges[:, 1], self.edges[:, 0]] = 1
        return a

    def get_degree_matrix(self, as_matrix: bool = True) -> np.ndarray:
        d = self.get_adjacency_matrix().sum(axis=0)
        return np.diag(d) if as_matrix else d

    def get_laplacian_matrix(self) -> np.ndarray:
        return self.get_degree_matrix() - self.get_adjacency_matrix()

    def get_normalized_adjacency_matrix(self, normalization: str = "row", add_self_connections: bool = False):
        if normalization not in _NORMALIZATIONS:
            # the reference raises while concatenating the message (TypeError for non-str input)
            raise ValueError("Unsupported normalization: " + normalization)
        adj = self.get_adjacency_matrix().astype(np.float64)
        if add_self_connections:
            adj = adj + np.eye(self.num_vertices)
        deg = adj.sum(axis=0)
        inv = np.zeros_like(deg)
        nz = deg > 0
        inv[nz] = 1.0 / (np.sqrt(deg[nz]) if normalization == "symmetric" else deg[nz])
        if normalization == "row":
            return inv[:, None] * adj
        if normalization == "column":
            return adj * inv[None, :]
        return inv[:, None] * adj * inv[None, :]

    def __str__(self) -> str:
        return f"|V| = {self.num_vertices}; |E| = {len(self.edges)}"


def get_k_adjacency(adj: np.ndarray, k: int, with_self: bool = False, self_factor: int = 1) -> np.ndarray:
    """Exact-k-hop adjacency (reference util/graph.py:176-184)."""
    eye = np.eye(len(adj), dtype=adj.dtype)
    if k == 0:
        return eye
    reach_k = np.minimum(np.linalg.matrix_power(adj + eye, k), 1)
    reach_km1 = np.minimum(np.linalg.matrix_power(adj + eye, k - 1), 1)
    out = reach_k - reach_km1
    if with_self:
        out = out + self_factor * eye
    return out
