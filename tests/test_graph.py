"""Adjacency construction: product host code (fusion_gcn_amd.util / datasets / fusion graph) and the
loop-level oracle, both against the reference's own outputs (tests/golden/adjacency.npz)."""
import numpy as np
import pytest

from fusion_gcn_amd.datasets.mmact import constants as mmact
from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
from fusion_gcn_amd.datasets.utd_mhad import constants as utd
from fusion_gcn_amd.models.mmargcn.fusion import get_skeleton_imu_fusion_graph
from fusion_gcn_amd.util import Graph, GraphPartitionStrategy
from fusion_gcn_amd.util.dynamic_import import import_dataset_constants
from oracle import graph_oracle


def _graphs():
    g = {"utd": Graph(utd.skeleton_edges, center_joint=utd.center_joint),
         "mmact": Graph(mmact.skeleton_edges, center_joint=mmact.center_joint),
         "ntu": Graph(ntu.skeleton_edges, center_joint=ntu.center_joint)}
    g["utd_imu2_center"] = get_skeleton_imu_fusion_graph(g["utd"], "append_center", 2)
    g["utd_imu2_center_ic"] = get_skeleton_imu_fusion_graph(g["utd"], "append_center", 2, interconnect_imu_joints=True)
    g["utd_imu2_right"] = get_skeleton_imu_fusion_graph(g["utd"], "append_right", 2, right_wrist_joint=10, right_hip_joint=16)
    g["utd_imu2_right_ic"] = get_skeleton_imu_fusion_graph(g["utd"], "append_right", 2, right_wrist_joint=10,
                                                           right_hip_joint=16, interconnect_imu_joints=True)
    g["mmact_imu4_center"] = get_skeleton_imu_fusion_graph(g["mmact"], "append_center", 4)
    g["mmact_imu4_center_ic"] = get_skeleton_imu_fusion_graph(g["mmact"], "append_center", 4, interconnect_imu_joints=True)
    g["ntu_imu2_center"] = get_skeleton_imu_fusion_graph(g["ntu"], "append_center", 2)
    return g


NAMES = sorted(_graphs())


@pytest.mark.parametrize("name", NAMES)
def test_product_adjacency_matches_reference(golden, name):
    ref = golden("adjacency.npz")
    g = _graphs()[name]
    np.testing.assert_array_equal(g.edges, ref[f"edges.{name}"])
    got = GraphPartitionStrategy().get_adjacency_matrix_array(g)
    assert got.dtype == np.float64 and got.shape == ref[name].shape
    np.testing.assert_array_equal(got, ref[name])          # bit-exact: 0, 1 and 1/deg entries only


@pytest.mark.parametrize("name", NAMES)
def test_oracle_adjacency_matches_reference(golden, name):
    ref = golden("adjacency.npz")
    got = graph_oracle.spatial_partition_stack(ref[f"edges.{name}"])
    np.testing.assert_array_equal(got, ref[name])


def test_oracle_imu_edges_match_reference(golden):
    ref = golden("adjacency.npz")
    e = graph_oracle.imu_fusion_edges(utd.skeleton_edges, 20, "append_right", 2, right_wrist_joint=10,
                                      right_hip_joint=16, interconnect=True)
    np.testing.assert_array_equal(np.array(e), ref["edges.utd_imu2_right_ic"])
    e = graph_oracle.imu_fusion_edges(mmact.skeleton_edges, 18, "append_center", 4, center_joint=1)
    np.testing.assert_array_equal(np.array(e), ref["edges.mmact_imu4_center"])


def test_stack_properties():
    a = GraphPartitionStrategy().get_adjacency_matrix_array(_graphs()["ntu"])
    assert [int((x != 0).sum()) for x in a] == [25, 24, 24]
    # column-normalised: every non-empty column sums to one
    for k in (1, 2):
        s = a[k].sum(axis=0)
        assert np.all((np.abs(s - 1) < 1e-15) | (s == 0))
    # centripetal subset: each non-centre joint receives exactly from its parent
    for child, parent in ntu.skeleton_edges:
        assert a[1][parent, child] == 1.0


def test_graph_api_edge_cases():
    g = Graph([(1, 0), (1, 0), (2, 1)])
    assert g.num_vertices == 3 and len(g.edges) == 2                       # duplicates collapse
    assert g.has_edge((1, 0)) and not g.has_edge((0, 1))
    assert g.as_directed().is_directed and g.as_directed().as_undirected().is_directed is False
    with pytest.raises(AssertionError):
        Graph([(0, 1)], num_vertices=1)
    with pytest.raises(AssertionError):
        Graph([(-1, 0)])
    # isolated vertex: zero-degree column stays all zero instead of NaN/inf
    iso = Graph([(1, 0)], num_vertices=4, is_directed=True).get_normalized_adjacency_matrix("column")
    assert np.isfinite(iso).all() and iso[:, 2].sum() == 0
    with pytest.raises(NotImplementedError):
        GraphPartitionStrategy("distance").get_adjacency_matrix_array(g)
    with pytest.raises(TypeError):
        GraphPartitionStrategy("uniform").get_adjacency_matrix_array(g)
    with pytest.raises(ValueError):
        get_skeleton_imu_fusion_graph(g, "nope", 1)
    # with_new_edges: undirected result, vertex count re-derived
    h = Graph([(1, 0)], num_vertices=5, is_directed=True).with_new_edges([(2, 1)])
    assert h.num_vertices == 3 and not h.is_directed


def test_dataset_constants_lookup():
    edges, cj, nc = import_dataset_constants("UTD-MHAD", ["skeleton_edges", "center_joint", "num_classes"])
    assert edges.shape == (19, 2) and cj == 1 and nc == 27
    assert import_dataset_constants("NTU-RGB-D", ["num_joints", "num_classes"]) == [25, 60]
    assert import_dataset_constants("MMAct", ["num_joints", "num_classes"]) == [18, 35]
