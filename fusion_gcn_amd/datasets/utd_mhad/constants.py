"""UTD-MHAD (Kinect v1, 20 joints, 1 body, 27 actions) — graph constants only.

Restates the tables the model layer reads from the reference's
datasets/utd_mhad/constants.py:63-112 (skeleton_joints, skeleton_edges oriented towards the
shoulder centre, center_joint, num_joints, num_classes).  Preprocessing constants are out of scope.
"""
from .._skeleton import edges_from_parents

skeleton_joints = [
    "head", "shoulder_center", "spine", "hip_center",
    "left_shoulder", "left_elbow", "left_wrist", "left_hand",
    "right_shoulder", "right_elbow", "right_wrist", "right_hand",
    "left_hip", "left_knee", "left_ankle", "left_foot",
    "right_hip", "right_knee", "right_ankle", "right_foot",
]

# child -> parent, every chain ends at shoulder_center (1)
_parent = {0: 1, 2: 1, 4: 1, 8: 1, 3: 2, 12: 3, 16: 3,
           5: 4, 6: 5, 7: 6, 9: 8, 10: 9, 11: 10,
           13: 12, 14: 13, 15: 14, 17: 16, 18: 17, 19: 18}
skeleton_edges = edges_from_parents(_parent)
center_joint = 1

num_joints = len(skeleton_joints)
num_classes = 27
num_subjects = 8
skeleton_max_sequence_length = 128
default_data_shape = (3, skeleton_max_sequence_length, 20, 1)  # (C, T, V, M)
