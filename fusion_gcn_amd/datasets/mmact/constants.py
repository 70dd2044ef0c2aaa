"""MMAct (OpenPose COCO-18 body, up to 2 bodies, 35 actions) — graph constants only.

Restates reference datasets/mmact/constants.py:60-113 (skeleton_joints, skeleton_edges towards the
neck, center_joint, num_joints, num_classes, num_subjects, num_views).
"""
from .._skeleton import edges_from_parents

skeleton_joints = [
    "head", "shoulder_center",
    "right_shoulder", "right_elbow", "right_hand",
    "left_shoulder", "left_elbow", "left_hand",
    "right_hip", "right_knee", "right_foot",
    "left_hip", "left_knee", "left_foot",
    "right_eye", "left_eye", "right_ear", "left_ear",
]

_parent = {0: 1, 2: 1, 5: 1, 8: 1, 11: 1,
           3: 2, 4: 3, 6: 5, 7: 6, 9: 8, 10: 9, 12: 11, 13: 12,
           14: 0, 15: 0, 16: 14, 17: 15}
skeleton_edges = edges_from_parents(_parent)
center_joint = 1

num_joints = len(skeleton_joints)
num_classes = 35
num_subjects = 20
num_views = 4
