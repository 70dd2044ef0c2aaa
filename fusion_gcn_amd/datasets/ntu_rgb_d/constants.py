"""NTU RGB+D (Kinect v2, 25 joints, 2 bodies, 60 actions) — graph constants only.

Restates reference datasets/ntu_rgb_d/constants.py:9,74-138 (default_data_shape, skeleton_joints,
skeleton bones oriented towards the spine joint 20, center_joint, num_joints, num_classes).
This is the (V=25, M=2, T=300) graph BASELINE.json's headline config runs on.
"""
from .._skeleton import edges_from_parents

skeleton_joints = [
    "spine_base", "spine_center", "neck", "head",
    "left_shoulder", "left_elbow", "left_wrist", "left_hand",
    "right_shoulder", "right_elbow", "right_wrist", "right_hand",
    "left_hip", "left_knee", "left_ankle", "left_foot",
    "right_hip", "right_knee", "right_ankle", "right_foot",
    "spine", "left_hand_tip", "left_thumb", "right_hand_tip", "right_thumb",
]

_parent = {0: 1, 1: 20, 2: 20, 3: 2, 4: 20, 5: 4, 6: 5, 7: 6,
           8: 20, 9: 8, 10: 9, 11: 10, 12: 0, 13: 12, 14: 13, 15: 14,
           16: 0, 17: 16, 18: 17, 19: 18, 21: 22, 22: 7, 23: 24, 24: 11}
skeleton_edges = edges_from_parents(_parent)
center_joint = 20

num_joints = len(skeleton_joints)
num_classes = 60
num_subjects = 40
max_sequence_length = 300
max_body_true = 2
default_data_shape = (3, 300, 25, 2)  # (C, T, V, M)
