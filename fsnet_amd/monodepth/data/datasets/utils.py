"""File readers of the KITTI raw layout with the reference's names (monodepth/data/datasets/utils.py:22-54):
PNG decode through PIL exactly like the reference's read_image, 16-bit ground-truth depth / 256, the MATLAB-devkit
pose file, and the camera-frame relative pose."""
import numpy as np
import scipy.io as sio
from PIL import Image


def read_image(path):
    """[H, W, 3] uint8 RGB (reference :22-30)"""
    return np.array(Image.open(path, 'r'))


def read_depth(path):
    """16-bit PNG / 256 -> float32 metres (reference :32-40 reads it with cv2.imread(path, -1))"""
    return np.array(np.asarray(Image.open(path, 'r'), dtype=np.float64) / 256.0, dtype=np.float32)


def read_pose_mat(path):
    """[N, 4, 4] imu-to-world poses written by the MATLAB devkit (reference :42-50)"""
    return sio.loadmat(path)['pose_mat']


def cam_relative_pose(T_imu2world_0, T_imu2world_1, T_imu2vel, T_vel2cam):
    """pose of camera frame 0 expressed in camera frame 1 (reference :53-54)"""
    return T_vel2cam @ T_imu2vel @ np.linalg.inv(T_imu2world_1) @ T_imu2world_0 @ np.linalg.inv(T_imu2vel) \
        @ np.linalg.inv(T_vel2cam)
