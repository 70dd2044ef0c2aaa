"""``mode: skeleton_imu_spatial_fusion`` — IMU modalities as extra skeleton joints, then AGCN.

Mirrors ``SkeletonImuSpatialFusionModel`` (reference torch_src/models/mmargcn/early_fusion_models.py:9-22).
``mode: skeleton_imu_channel_fusion`` -- ``SkeletonImuChannelFusionModel`` (:25-45): the IMU signals of a frame are broadcast to
every joint as extra input channels.  The RGB early-fusion variants of that file wrap image encoders and are out of scope
(SURVEY.md §2 row 10).
"""
import torch.nn as nn

from . import agcn
from .fusion import get_fusion, get_skeleton_imu_fusion_graph


class SkeletonImuSpatialFusionModel(nn.Module):
    def __init__(self, data_shape, num_classes: int, graph, **kwargs):
        super().__init__()
        num_layers = kwargs.get("num_layers", 10)
        skeleton_imu_graph = get_skeleton_imu_fusion_graph(graph, **kwargs)
        self.agcn = agcn.Model(data_shape["skeleton"], num_classes, skeleton_imu_graph, num_layers=num_layers,
                               without_fc=kwargs.get("without_fc", False),
                               static_adjacency=kwargs.get("static_adjacency", False),
                               fused_spatial=kwargs.get("fused_spatial", True))

    def forward(self, x):
        return self.agcn(x)


class SkeletonImuChannelFusionModel(nn.Module):
    """Extend the skeleton data's channels by broadcasting the IMU data to each node (inertial frames == skeleton frames)."""

    def __init__(self, data_shape, num_classes: int, graph, **kwargs):
        super().__init__()
        num_layers = kwargs.get("num_layers", 10)
        shape = list(data_shape["skeleton"])
        shape[-1] += data_shape["inertial"][-1]
        self.fusion = get_fusion("concatenate", concatenate_dim=-1)
        self.agcn = agcn.Model(tuple(shape), num_classes, graph, num_layers=num_layers, without_fc=kwargs.get("without_fc", False))

    def forward(self, x):
        skeleton_data, imu_data = x["skeleton"], x["inertial"]
        imu_data = imu_data.unsqueeze(1).unsqueeze(3)              # (N, 1, T, 1, S): bodies and joints
        imu_data = imu_data.expand(-1, skeleton_data.shape[1], -1, skeleton_data.shape[3], -1)
        return self.agcn(self.fusion.combine(skeleton_data, imu_data))
