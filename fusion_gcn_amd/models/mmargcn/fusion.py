"""Skeleton + IMU fusion graph (host side, build time).

Mirrors ``get_skeleton_imu_fusion_graph`` of the reference (torch_src/models/mmargcn/fusion.py:65-89):
IMU modalities become ``num_imu_joints`` extra vertices V..V+n-1 attached either to the skeleton's
centre joint ("append_center") or to right wrist + right hip ("append_right"), optionally pairwise
inter-connected.  The ``Fusion`` combiner classes of that file (late / feature fusion) are out of the
hot-path scope.
"""
from ...util.graph import Graph


def get_skeleton_imu_fusion_graph(skeleton_graph: Graph, imu_enhanced_mode: str, num_imu_joints: int, **kwargs) -> Graph:
    first = skeleton_graph.num_vertices
    imu = range(first, first + num_imu_joints)
    if imu_enhanced_mode == "append_center":
        anchor = kwargs.get("center_joint", skeleton_graph.center_joint)
        new_edges = [(j, anchor) for j in imu]
    elif imu_enhanced_mode == "append_right":
        wrist, hip = kwargs["right_wrist_joint"], kwargs["right_hip_joint"]
        new_edges = [(j, a) for j in imu for a in (wrist, hip)]
    else:
        raise ValueError("Unsupported imu_enhanced_mode: " + imu_enhanced_mode)
    if kwargs.get("interconnect_imu_joints", False):
        new_edges += [(a, b) for a in imu for b in imu if a < b]
    return skeleton_graph.with_new_edges(new_edges)
