"""Pose bookkeeping of the mirror augmentation (reference vision_base/data/augmentations/utils.py:4-20)."""
import numpy as np
from scipy.spatial.transform import Rotation


def flip_relative_pose(pose, axis_num=0):
    """Relative camera pose after the world is mirrored along `axis_num`: the Euler angles about the other two axes
    change sign, as does the translation along the axis."""
    angles = Rotation.from_matrix(pose[0:3, 0:3]).as_euler('xyz')
    signs = np.array([1.0 if i == axis_num else -1.0 for i in range(3)])
    flipped = np.eye(4, dtype=np.float32)
    flipped[0:3, 0:3] = Rotation.from_euler('xyz', angles * signs).as_matrix()
    flipped[0:3, 3] = pose[0:3, 3]
    flipped[axis_num, 3] *= -1
    return flipped
