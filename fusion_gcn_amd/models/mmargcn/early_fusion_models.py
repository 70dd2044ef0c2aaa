"""``mode: skeleton_imu_spatial_fusion`` — IMU modalities as extra skeleton joints, then AGCN.

Mirrors ``SkeletonImuSpatialFusionModel`` (reference torch_src/models/mmargcn/early_fusion_models.py:9-22).
The other early-fusion variants of that file wrap RGB / channel-fusion encoders and are out of the hot-path scope
(SURVEY.md §2 row 10).
"""
import torch.nn as nn

from . import agcn
from .fusion import get_skeleton_imu_fusion_graph


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
