"""``model: mmargcn`` — mode dispatcher (reference torch_src/models/mmargcn/mmargcn.py:9-57).

Only the modes on the AGCN / ST-GCN hot path -- plus ``imu_gcn`` and ``skeleton_imu_gcn_late_fusion`` --
are backed by HIP kernels; the reference's other 12 modes (RGB encoders, signal images, late fusion) are out of scope
 and raise with a clear message instead of silently running something else.
"""
import torch.nn as nn

from . import early_fusion_models, imu_feature_models, late_fusion_models

_OUT_OF_SCOPE = (
    "rgb_patch_features", "rgb_patch_groups_features", "rgb_encoder_model", "rgb_r2p1d",
    "imu_signal_image", "skeleton_rgb_patch_features_early_fusion", "skeleton_rgb_encoding_early_fusion",
    "skeleton_rgb_encoding_r2p1d_early_fusion", "skeleton_rgb_r2p1d_late_fusion",
    "skeleton_imu_rgb_cnn_encoder_early_fusion",
    "skeleton_imu_rgb_patch_features_early_fusion", "skeleton_imu_rgb_r2p1d_early_fusion",
)


class Model(nn.Module):
    def __init__(self, data_shape, num_classes: int, graph, mode: str, **kwargs):
        super().__init__()
        modes = {"skeleton_imu_spatial_fusion": early_fusion_models.SkeletonImuSpatialFusionModel,
                 "skeleton_imu_channel_fusion": early_fusion_models.SkeletonImuChannelFusionModel,
                 "imu_gcn": imu_feature_models.ImuGCN,
                 "skeleton_imu_gcn_late_fusion": late_fusion_models.SkeletonImuGCN}
        if mode in _OUT_OF_SCOPE:
            raise NotImplementedError(f"mode {mode!r} is outside the MI355X hot-path build (AGCN / ST-GCN blocks only)")
        if mode not in modes:
            raise ValueError("Unsupported mode: " + mode)
        self._model = modes[mode](data_shape, num_classes, graph=graph, **kwargs)

    def forward(self, x):
        return self._model(x)
