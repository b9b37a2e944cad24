"""Synthetic 3-frame triplet dataset obeying the batch contract of the reference's data layer
(tuple keys, NCHW fp32 images in [0,1] + ImageNet-normalised copies, P2 [3,4], relative
poses, float64 patched_mask).  Stands in for KittiDepthMonoDataset (needs cv2 + data on disk) when driving
the training loop without a dataset."""
import numpy as np
import torch
from torch.utils.data import Dataset


def synthetic_mei_calib(H, W, variant=0):
    """KITTI-360-like left / right fisheye calibration (Mei unified camera model: mirror xi, radial k1 k2, gamma, u0 v0)
    rescaled from the 1400 x 1400 sensor to H x W — the `P2` / `calib_meta` entries the KITTI-360 fisheye reader hands to
    FishEyeDecoder (configs/kitti360_fisheye_example; monodepth2_decoder.py:355-411)."""
    s = H / 1400.0
    if variant == 0:
        xi, k1, k2, g1, g2, u0, v0 = 2.2134047, 0.016798, 1.6548, 1336.3, 1335.8, 716.94, 705.76
    else:
        xi, k1, k2, g1, g2, u0, v0 = 2.5535139, 0.049134, 4.5014, 1485.4, 1484.9, 698.88, 698.14
    calib = {"distortion_parameters": {"k1": k1, "k2": k2}, "mirror_parameters": {"xi": xi}}
    P = np.array([[g1 * s, 0, u0 * s * W / H, 0], [0, g2 * s, v0 * s, 0], [0, 0, 1, 0]], dtype=np.float32)
    return P, calib


class SyntheticTripletDataset(Dataset):
    def __init__(self, size=256, height=192, width=640, frame_idxs=(0, 1, -1), seed=0, fisheye=False, **kwargs):
        self.size, self.H, self.W, self.frames, self.seed = size, height, width, list(frame_idxs), seed
        self.fisheye = fisheye
        self.mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        self.std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        H, W = self.H, self.W
        ys = torch.linspace(0, 1, H).view(1, H, 1)
        xs = torch.linspace(0, 1, W).view(1, 1, W)
        ph = torch.rand(3, 1, 1, generator=g) * 6.28
        fr = 3 + torch.rand(3, 1, 1, generator=g) * 9
        out = {}
        for f in self.frames:
            sh = 3.0 * f / W
            img = 0.5 + 0.25 * torch.sin(fr * 6.28 * (xs + sh) + ph) * torch.cos(fr * 3.1 * ys + 0.5 * ph) \
                + 0.2 * torch.sin(37.0 * (xs + sh) * ys + ph) + 0.03 * torch.rand(3, H, W, generator=g)
            img = img.clamp(0, 1).float()
            out[("original_image", f)] = img
            out[("image", f)] = (img - self.mean) / self.std
        P2 = np.zeros((3, 4), dtype=np.float32)
        P2[0, 0], P2[0, 2], P2[1, 1], P2[1, 2], P2[2, 2] = 0.58 * W, 0.5 * W, 1.92 * H, 0.5 * H, 1
        out["P2"] = P2
        if self.fisheye:       # two cameras alternate within a batch, like the left / right fisheye of a KITTI-360 drive
            out["P2"], out["calib_meta"] = synthetic_mei_calib(H, W, i % 2)
        for f in self.frames[1:]:
            T = np.eye(4, dtype=np.float32)
            T[0, 3], T[2, 3] = 0.01, (-0.8 if f > 0 else 0.8)
            if self.fisheye:   # sideways-looking camera: the vehicle's motion is mostly along the image x axis
                T[0, 3], T[2, 3] = (0.6 if f > 0 else -0.6), 0.03
            out[("relative_pose", f)] = T
        out["patched_mask"] = np.ones((H, W), dtype=np.float64)
        return out
