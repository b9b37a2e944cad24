"""Shared set-up of the input-pipeline tests: the golden samples of tests/golden/augment.npz are regenerated from
their seeds (frames, P2, poses) exactly as tools/gen_golden.py::gen_augment made them."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz")
FRAME_IDXS = [0, 1, -1]


def golden():
    return np.load(GOLD)


def sample_inputs(g, n):
    from scipy.spatial.transform import Rotation as R
    H, W = int(g["H"]), int(g["W"])
    fr = np.random.RandomState(int(g["frame_seed"]) + n)
    frames = [fr.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in FRAME_IDXS]
    P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)
    poses = []
    for _ in range(2):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R.from_euler('xyz', fr.uniform(-0.05, 0.05, 3)).as_matrix()
        T[:3, 3] = fr.uniform(-1, 1, 3)
        poses.append(T)
    return frames, P2, poses


def sample_dict(frames, P2, poses):
    data = {}
    for i, f in zip(FRAME_IDXS, frames):
        data[('image', i)] = f.copy()
        data[('original_image', i)] = f.copy()
    data['patched_mask'] = np.ones(frames[0].shape[:2])
    data['P2'] = P2.copy()
    for i, T in zip(FRAME_IDXS[1:], poses):
        data[('relative_pose', i)] = T.copy()
    return data


def pipeline_cfg(g, prefix='fsnet_amd.vision_base.data.augmentations.augmentations',
                 builder='fsnet_amd.vision_base.utils.builder'):
    """configs/kitti_wpose_example:129-155 with the golden's sizes and seeds"""
    out_h, out_w = int(g["out_h"]), int(g["out_w"])
    resize_keys = [('image', i) for i in FRAME_IDXS] + [('original_image', i) for i in FRAME_IDXS]
    colour_keys = [('image', i) for i in FRAME_IDXS]
    return dict(name=builder + '.Sequential', cfg_list=[
        dict(name=prefix + '.ConvertToFloat'),
        dict(name=prefix + '.RandomWarpAffine', output_w=out_w, output_h=out_h, random_seed=int(g["seed_warp"])),
        dict(name=prefix + '.RandomMirror', mirror_prob=0.5,
             pose_axis_pairs=[(("relative_pose", i), 0) for i in FRAME_IDXS[1:]]),
        dict(name=builder + '.Shuffle', cfg_list=[
            dict(name=prefix + '.RandomBrightness', distort_prob=1.0, random_seed=int(g["seed_bright"])),
            dict(name=prefix + '.RandomContrast', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=int(g["seed_contrast"])),
            dict(name=builder + '.Sequential', cfg_list=[
                dict(name=prefix + '.ConvertColor', transform='HSV'),
                dict(name=prefix + '.RandomSaturation', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=int(g["seed_sat"])),
                dict(name=prefix + '.ConvertColor', current='HSV', transform='RGB')])],
             image_keys=colour_keys),
        dict(name=prefix + '.Normalize', mean=g["mean"], stds=g["std"], image_keys=colour_keys),
        dict(name=prefix + '.Normalize', mean=np.array([0, 0, 0]), stds=np.array([1, 1, 1]),
             image_keys=[('original_image', i) for i in FRAME_IDXS]),
        dict(name=prefix + '.ConvertToTensor')],
        image_keys=resize_keys, calib_keys=['P2'], gt_image_keys=['patched_mask'])


VAL_CASES = (("stretch", dict(preserve_aspect_ratio=False)), ("pad1", dict(preserve_aspect_ratio=True, force_pad=True)),
             ("pad0", dict(preserve_aspect_ratio=True, force_pad=True)), ("crop1", dict(preserve_aspect_ratio=True, force_pad=False)))


def val_pipeline(g, kw, prefix='fsnet_amd.vision_base.data.augmentations.augmentations',
                 builder='fsnet_amd.vision_base.utils.builder'):
    """configs/kitti_wpose_example:156-166 at the golden's 48x160"""
    from fsnet_amd.vision_base.utils.builder import build
    return build(name=builder + '.Sequential', cfg_list=[
        dict(name=prefix + '.ConvertToFloat'), dict(name=prefix + '.Resize', size=(48, 160), **kw),
        dict(name=prefix + '.Normalize', mean=g['mean'], stds=g['std']), dict(name=prefix + '.ConvertToTensor')],
        image_keys=[('image', 0)], calib_keys=['P2'])


def val_frame(g, tag):
    shp = tuple(int(v) for v in g['val_%s_shape' % tag])
    fr = np.random.RandomState(int(g['val_%s_seed' % tag]))
    return fr.randint(0, 256, size=shp + (3,)).astype(np.uint8)


VAL_P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)


# ---- Resize-based training chain (configs/multi_dataset_example:178-205; tests/golden/augment_resize.npz) ----------
GOLD_RESIZE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_resize.npz")


def resize_sample_inputs(g, n):
    from scipy.spatial.transform import Rotation as R
    H, W = (int(v) for v in g["shapes"][n])
    fr = np.random.RandomState(int(g["frame_seed"]) + n)
    frames = [fr.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in FRAME_IDXS]
    P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)
    poses = []
    for _ in range(2):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R.from_euler('xyz', fr.uniform(-0.05, 0.05, 3)).as_matrix()
        T[:3, 3] = fr.uniform(-1, 1, 3)
        poses.append(T)
    return frames, P2, poses


def resize_pipeline_cfg(g, prefix='fsnet_amd.vision_base.data.augmentations.augmentations',
                        builder='fsnet_amd.vision_base.utils.builder'):
    size = tuple(int(v) for v in g["size"])
    resize_keys = [('image', i) for i in FRAME_IDXS] + [('original_image', i) for i in FRAME_IDXS]
    colour_keys = [('image', i) for i in FRAME_IDXS]
    return dict(name=builder + '.Sequential', cfg_list=[
        dict(name=prefix + '.ConvertToFloat'),
        dict(name=prefix + '.Resize', size=size, preserve_aspect_ratio=True, force_pad=True),
        dict(name=builder + '.Shuffle', cfg_list=[
            dict(name=prefix + '.RandomBrightness', distort_prob=1.0, random_seed=int(g["seed_bright"])),
            dict(name=prefix + '.RandomContrast', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=int(g["seed_contrast"])),
            dict(name=builder + '.Sequential', cfg_list=[
                dict(name=prefix + '.ConvertColor', transform='HSV'),
                dict(name=prefix + '.RandomSaturation', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=int(g["seed_sat"])),
                dict(name=prefix + '.ConvertColor', current='HSV', transform='RGB')])],
             image_keys=colour_keys),
        dict(name=prefix + '.RandomMirror', mirror_prob=0.5,
             pose_axis_pairs=[(("relative_pose", i), 0) for i in FRAME_IDXS[1:]]),
        dict(name=prefix + '.Normalize', mean=g["mean"], stds=g["std"], image_keys=colour_keys),
        dict(name=prefix + '.Normalize', mean=np.array([0, 0, 0]), stds=np.array([1, 1, 1]),
             image_keys=[('original_image', i) for i in FRAME_IDXS]),
        dict(name=prefix + '.ConvertToTensor')],
        image_keys=resize_keys, calib_keys=['P2'], gt_image_keys=['patched_mask'])
