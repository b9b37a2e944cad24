"""A tiny KITTI-raw tree on disk (one date, one drive, both cameras, PNG frames, calibration files, oxts/pose.mat,
a split file) generated from a seed — shared by tools/gen_golden.py::gen_kitti_dataset (which runs the REAL
KittiDepthMonoDataset over it) and the tests."""
import os

import numpy as np

DATE, DRIVE = "2011_09_26", "2011_09_26_drive_0001_sync"
H, W, NFRAMES = 24, 80, 8


def make_tree(root, seed=5, H=H, W=W):
    import scipy.io as sio
    from PIL import Image
    from scipy.spatial.transform import Rotation as R
    rng = np.random.RandomState(seed)
    raw = os.path.join(root, "raw")
    date = os.path.join(raw, DATE)
    os.makedirs(date, exist_ok=True)
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    P3 = P2.copy(); P3[0, 3] = -339.5242
    with open(os.path.join(date, "calib_cam_to_cam.txt"), "w") as f:
        f.write("calib_time: 09-Jan-2012 13:57:47\n")
        f.write("P_rect_02: " + " ".join("%.6e" % v for v in P2.flatten()) + "\n")
        f.write("P_rect_03: " + " ".join("%.6e" % v for v in P3.flatten()) + "\n")
    Rv = R.from_euler("xyz", [1.55, -0.01, 1.57]).as_matrix()
    with open(os.path.join(date, "calib_velo_to_cam.txt"), "w") as f:
        f.write("calib_time: 15-Mar-2012 11:37:16\n")
        f.write("R: " + " ".join("%.6e" % v for v in Rv.flatten()) + "\n")
        f.write("T: -4.069766e-03 -7.631618e-02 -2.717806e-01\n")
    Ri = R.from_euler("xyz", [0.002, -0.001, 0.015]).as_matrix()
    with open(os.path.join(date, "calib_imu_to_velo.txt"), "w") as f:
        f.write("calib_time: 25-May-2012 16:47:16\n")
        f.write("R: " + " ".join("%.6e" % v for v in Ri.flatten()) + "\n")
        f.write("T: -8.086759e-01 3.195559e-01 -7.997231e-01\n")
    drive = os.path.join(date, DRIVE)
    for cam in ("image_02", "image_03"):
        os.makedirs(os.path.join(drive, cam, "data"), exist_ok=True)
        for i in range(NFRAMES):
            Image.fromarray(rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)).save(
                os.path.join(drive, cam, "data", "%010d.png" % i))
    poses = np.zeros((NFRAMES, 4, 4))
    x = 0.0
    for i in range(NFRAMES):
        poses[i] = np.eye(4)
        poses[i][:3, :3] = R.from_euler("xyz", rng.uniform(-0.01, 0.01, 3)).as_matrix()
        x += 0.0 if i in (5, 6) else 0.8          # frames 4..6 stand still: the static filter must drop index 5
        poses[i][:3, 3] = (x, 0.02 * i, 0.0)
    os.makedirs(os.path.join(drive, "oxts"), exist_ok=True)
    sio.savemat(os.path.join(drive, "oxts", "pose.mat"), {"pose_mat": poses})
    split = os.path.join(root, "split.txt")
    with open(split, "w") as f:
        for i, side in ((1, "l"), (2, "r"), (3, "l"), (5, "l"), (6, "r")):
            f.write("%s/%s %d %s\n" % (DATE, DRIVE, i, side))
    return raw, split


def dataset_cfg(raw, split, prefix):
    """ConvertToFloat + Normalize + ConvertToTensor only (no cv2 calls: the reference class runs unshimmed)"""
    aug = prefix + 'vision_base.data.augmentations.augmentations'
    frame_idxs = [0, 1, -1]
    return dict(raw_path=raw, split_file=split, frame_idxs=frame_idxs, is_filter_static=True,
                augmentation=dict(name=prefix + 'vision_base.utils.builder.Sequential', cfg_list=[
                    dict(name=aug + '.ConvertToFloat'),
                    dict(name=aug + '.Normalize', mean=np.array([0.485, 0.456, 0.406]), stds=np.array([0.229, 0.224, 0.225]),
                         image_keys=[('image', i) for i in frame_idxs]),
                    dict(name=aug + '.Normalize', mean=np.array([0, 0, 0]), stds=np.array([1, 1, 1]),
                         image_keys=[('original_image', i) for i in frame_idxs]),
                    dict(name=aug + '.ConvertToTensor')],
                    image_keys=[('image', i) for i in frame_idxs] + [('original_image', i) for i in frame_idxs],
                    calib_keys=['P2'], gt_image_keys=['patched_mask']))


def add_velodyne(raw, seed=9, H=H, W=W, npts=5000):
    """velodyne scans + the calibration entries the ground-truth export reads (S_rect_02, R_rect_00) for the frames of
    make_tree's drive.  Points are drawn in the image (a margin outside included) and carried back to the velodyne
    frame, so that many share a pixel of the tiny image — duplicates and the export's index collisions — and a
    tenth of them sit behind the sensor."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.RandomState(seed)
    date = os.path.join(raw, DATE)
    Rr = R.from_euler("xyz", [0.003, -0.002, 0.001]).as_matrix()
    # (this tree's own calibration: a camera that looks along the velodyne's forward axis and sees the tiny frame)
    P2 = np.array([[60.0, 0, W / 2.0, 3.0], [0, 60.0, H / 2.0, 0.1], [0, 0, 1, 0.002]])
    with open(os.path.join(date, "calib_cam_to_cam.txt"), "w") as f:
        f.write("calib_time: 09-Jan-2012 13:57:47\n")
        f.write("S_rect_02: %.6e %.6e\n" % (W, H))
        f.write("R_rect_00: " + " ".join("%.6e" % v for v in Rr.flatten()) + "\n")
        f.write("P_rect_02: " + " ".join("%.6e" % v for v in P2.flatten()) + "\n")
    Rv = np.array([[0.0, -1, 0], [0, 0, -1], [1, 0, 0]]) @ R.from_euler("xyz", [0.01, -0.02, 0.015]).as_matrix()
    with open(os.path.join(date, "calib_velo_to_cam.txt"), "w") as f:
        f.write("calib_time: 15-Mar-2012 11:37:16\n")
        f.write("R: " + " ".join("%.6e" % v for v in Rv.flatten()) + "\n")
        f.write("T: -4.069766e-03 -7.631618e-02 -2.717806e-01\n")
    from fsnet_amd.monodepth.networks.utils.monodepth_utils import read_calib_file
    c2c = read_calib_file(os.path.join(date, "calib_cam_to_cam.txt"))
    v2c = read_calib_file(os.path.join(date, "calib_velo_to_cam.txt"))
    P = c2c["P_rect_02"].reshape(3, 4)
    T = np.eye(4); T[:3, :3] = v2c["R"].reshape(3, 3); T[:3, 3] = v2c["T"]
    R4 = np.eye(4); R4[:3, :3] = c2c["R_rect_00"].reshape(3, 3)
    to_velo = np.linalg.inv(R4 @ T)
    d = os.path.join(date, DRIVE, "velodyne_points", "data")
    os.makedirs(d, exist_ok=True)
    for i in range(NFRAMES):
        u = rng.uniform(-4, W + 6, npts); v = rng.uniform(-4, H + 6, npts); z = rng.uniform(2, 60, npts)
        X = (u * (z + P[2, 3]) - P[0, 2] * z - P[0, 3]) / P[0, 0]
        Y = (v * (z + P[2, 3]) - P[1, 2] * z - P[1, 3]) / P[1, 1]
        cam = np.stack([X, Y, z, np.ones(npts)], 1)
        pts = (to_velo @ cam.T).T.astype(np.float32)
        back = rng.rand(npts) < 0.1
        pts[back, 0] = -np.abs(pts[back, 0])
        pts[:, 3] = rng.uniform(0, 1, npts).astype(np.float32)
        pts.tofile(os.path.join(d, "%010d.bin" % i))
