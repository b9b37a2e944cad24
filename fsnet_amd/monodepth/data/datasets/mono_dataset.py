"""KITTI raw monocular triplet dataset with the reference's class name, constructor keys and sample contract
(monodepth/data/datasets/mono_dataset.py:17-250): split file -> (folder, index, side), per-date calibration
(P_rect_02/03, velo->cam, imu->velo), oxts poses -> ('relative_pose', f), raw uint8 frames under
('image', f) / ('original_image', f), float64 patched_mask of ones, P2 / original_P2 — then the configured
augmentation.  With the mirrored augmentation classes the frames stay uint8 and the pixel work runs on the device
(vision_base/data/augmentations: DeviceAugment); the dataset itself only decodes PNGs and does 4x4 algebra.

Not mirrored: precomputed motion masks / optical flow (cv2.imread of side products no shipped config enables)."""
import os
from copy import deepcopy

import numpy as np
import torch.utils.data

from fsnet_amd.monodepth.data.datasets.utils import cam_relative_pose, read_depth, read_image, read_pose_mat
from fsnet_amd.vision_base.utils.builder import build
from fsnet_amd.vision_base.utils.utils import EasyDict


def read_P23_from_sequence(file):
    """P_rect_02 / P_rect_03 of calib_cam_to_cam.txt (reference :25-42)"""
    P2 = P3 = None
    with open(file, 'r') as f:
        for line in f.readlines():
            data = line.split(" ")
            if line.startswith("P_rect_02"):
                P2 = np.reshape(np.array([float(x) for x in data[1:13]]), [3, 4])
            if line.startswith("P_rect_03"):
                P3 = np.reshape(np.array([float(x) for x in data[1:13]]), [3, 4])
    assert P2 is not None, f"can not find P2 in file {file}"
    assert P3 is not None, f"can not find P3 in file {file}"
    return P2, P3


def _read_RT(file, r_key, t_key):
    R = t = None
    with open(file, 'r') as f:
        for line in f.readlines():
            data = line.split(" ")
            if line.startswith(r_key):
                R = np.reshape(np.array([float(x) for x in data[1:10]]), [3, 3])
            if line.startswith(t_key):
                t = np.reshape(np.array([float(x) for x in data[1:4]]), [3, 1])
    assert R is not None, f"can not find R in file {file}"
    assert t is not None, f"can not find T in file {file}"
    T = np.eye(4)
    T[0:3, 0:3] = R
    T[0:3, 3:4] = t
    return T


def read_imu2velo(file):
    """calib_imu_to_velo.txt (reference :44-63: lines starting with "R" / "T")"""
    return _read_RT(file, "R", "T")


def read_T_from_sequence(file):
    """calib_velo_to_cam.txt (reference :65-87: lines "R:" / "T:")"""
    return _read_RT(file, "R:", "T:")


def read_split_file(file):
    """lines "<date>/<drive> <index> <l|r>" (reference :89-107)"""
    imdb = []
    with open(file, 'r') as f:
        for line in f.readlines():
            parts = line.strip().split()
            if not parts:
                continue
            imdb.append(dict(folder=parts[0], index=int(parts[1]), side=parts[2], datetime=parts[0].split("/")[0]))
    return imdb


class KittiDepthMonoDataset(torch.utils.data.Dataset):
    def __init__(self, **data_cfg):
        data_cfg = EasyDict(data_cfg)
        super().__init__()
        self.raw_path = data_cfg.raw_path
        self.depth_path = getattr(data_cfg, 'depth_path', None)
        self.frame_idxs = data_cfg.frame_idxs
        self.imdb = read_split_file(data_cfg.split_file)
        self.meta_dict = {}
        for date_time in sorted(os.listdir(self.raw_path)):
            folder_path = os.path.join(self.raw_path, date_time)
            if not os.path.isdir(folder_path):
                continue
            P2, P3 = read_P23_from_sequence(os.path.join(folder_path, "calib_cam_to_cam.txt"))
            self.meta_dict[date_time] = dict(
                P2=P2, P3=P3, T_vel2cam=read_T_from_sequence(os.path.join(folder_path, "calib_velo_to_cam.txt")),
                T_imu2vel=read_imu2velo(os.path.join(folder_path, "calib_imu_to_velo.txt")))
        self.pose_dict = {key: read_pose_mat(os.path.join(self.raw_path, key, 'oxts', 'pose.mat'))
                          for key in set(obj['folder'] for obj in self.imdb)}
        if getattr(data_cfg, 'is_motion_mask', False) or getattr(data_cfg, 'is_precompute_flow', False):
            raise NotImplementedError("precomputed motion masks / flow are not part of the mirrored data path")
        self.is_filter_static = getattr(data_cfg, 'is_filter_static', True)
        if self.is_filter_static:
            self.imdb = self._filter_static_indexes()
        self.transform = build(**data_cfg.augmentation)

    def _relative_poses(self, folder, index, datetime):
        imu2world = self.get_pose(folder, [index + idx for idx in self.frame_idxs])
        T_imu2vel, T_vel2cam = self.meta_dict[datetime]['T_imu2vel'], self.meta_dict[datetime]['T_vel2cam']
        return [cam_relative_pose(imu2world[0], imu2world[i + 1], T_imu2vel, T_vel2cam).astype(np.float32)
                for i in range(len(self.frame_idxs) - 1)]

    def _filter_static_indexes(self):
        """drop samples whose camera moved < 3 cm to either neighbour frame (reference :151-170)"""
        return [obj for obj in self.imdb
                if not any(np.linalg.norm(pose[0:3, 3]) < 0.03
                           for pose in self._relative_poses(obj['folder'], obj['index'], obj['datetime']))]

    def __getitem__(self, i):
        obj = self.imdb[i]
        folder, index, side, datetime = obj['folder'], obj['index'], obj['side'], obj['datetime']
        data = dict()
        for idx in self.frame_idxs:
            data[("image", idx)] = self.get_color(folder, index + idx, side)
            data[('original_image', idx)] = data[('image', idx)].copy()
        h, w, _ = data[("image", 0)].shape
        data["patched_mask"] = np.ones([h, w])
        for idx, pose in zip(self.frame_idxs[1:], self._relative_poses(folder, index, datetime)):
            data[('relative_pose', idx)] = pose
        data['P2'] = self.meta_dict[datetime][{"l": "P2", "r": "P3"}[side]]
        data['original_P2'] = data['P2'].copy()
        if self.depth_path is not None:
            data[('sparse_depth', 0)] = self.get_depth(folder, index, side)
        return self.transform(deepcopy(data))

    def __len__(self):
        return len(self.imdb)

    def get_color(self, folder, frame_index, side):
        camera_folder = {"l": "image_02", "r": "image_03"}[side]
        return read_image(os.path.join(self.raw_path, folder, camera_folder, 'data', '%010d.png' % frame_index))

    def get_depth(self, folder, frame_index, side):
        camera_folder = {"l": "image_02", "r": "image_03"}[side]
        return read_depth(os.path.join(self.depth_path, folder.split('/')[1], 'proj_depth', 'groundtruth', camera_folder,
                                       "%010d.png" % frame_index))

    def get_pose(self, folder, frame_indexes, *args, **kwargs):
        return self.pose_dict[folder][frame_indexes, :, :]
