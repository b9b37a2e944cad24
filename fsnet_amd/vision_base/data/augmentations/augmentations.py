"""Training input pipeline with the pixel work on the device.

Same class names, keyword arguments, random-number streams and dict keys as the reference's
vision_base/data/augmentations/augmentations.py, so configs/kitti_wpose_example:129-155 runs with its `name=`
strings repointed here.  The difference is WHERE pixels are touched: the reference's transforms run cv2 / numpy on
float32 HWC images inside DataLoader workers (3 x 375x1242 frames per sample — at a GPU step of a few ms four
workers cannot keep up); these transforms only draw their random numbers, do the O(1) bookkeeping the reference
does (P2, relative poses) and append to a per-sample *plan*.  Frames stay uint8.  `DeviceAugment.collate` (or
`materialize`) uploads the raw bytes of a batch once and one `fs_augment_frames` launch produces every
('image', i) / ('original_image', i) tensor and the patched mask on the GPU.

A transform that is given float images without a plan to extend (i.e. used outside this flow) raises: there is no
CPU pixel path here.
"""
import ctypes as C

import numpy as np
from numpy import random
import torch

from .utils import flip_relative_pose

PLAN = '_fs_aug_plan'
OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION = 0, 1, 2


def _plan(data):
    p = data.get(PLAN)
    if p is None:
        p = data[PLAN] = {"warp": None, "mirror": False, "ops": [], "normalize": {}, "hsv": False}
    return p


def _frame_shape(data, key):
    img = data[key]
    if not (isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.ndim == 3):
        raise TypeError("fsnet_amd augmentations plan device work over raw uint8 HWC frames; %r is %s" % (
            key, getattr(img, "dtype", type(img))))
    return img.shape


class EmptyAug(object):
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, data):
        return data


class ExtractData(object):
    """reference :30-47"""
    def __init__(self, extract_keys=[], mapped_keys={}):
        self.extract_keys, self.mapped_keys = extract_keys, mapped_keys

    def __call__(self, data):
        out = {k: data[k] for k in self.extract_keys}
        for k, new in self.mapped_keys.items():
            out[new] = data[k]
        if PLAN in data:
            out[PLAN] = data[PLAN]
        return out


class ConvertToFloat(object):
    """reference :50-59.  Frames stay uint8 here; the kernel converts while it samples."""
    def __init__(self, image_keys=['image'], **kwargs):
        self.image_keys = image_keys

    def __call__(self, data):
        for key in self.image_keys:
            _frame_shape(data, key)
        _plan(data)
        return data


class Resize(object):
    """reference :112-198: cv2.resize to (h, w), then crop / zero-pad to `size`; P2 rows scale.  The validation input
    of every shipped config, and the first stage of the training input of the Resize-based ones
    (configs/multi_dataset_example:183, nusc / kitti360_fisheye examples), where colour ops, RandomMirror and the
    Normalizes follow and `gt_image_keys` carries the patched mask (cv2.INTER_NEAREST)."""
    def __init__(self, size, preserve_aspect_ratio=True, force_pad=True, image_keys=['image'], calib_keys=[],
                 gt_image_keys=[], **kwargs):
        if any(k != 'patched_mask' for k in gt_image_keys):
            raise NotImplementedError("the only ground-truth image on the device path is 'patched_mask'")
        self.size, self.preserve_aspect_ratio, self.force_pad = size, preserve_aspect_ratio, force_pad
        self.image_keys, self.calib_keys, self.gt_image_keys = image_keys, calib_keys, list(gt_image_keys)

    def __call__(self, data):
        shape = _frame_shape(data, self.image_keys[0])
        plan = _plan(data)
        if plan["warp"] is not None or plan.get("resize") is not None or plan["ops"] or plan["mirror"]:
            raise NotImplementedError("Resize is the first (and only geometric) stage of a device pipeline")
        for key in self.gt_image_keys:
            if key in data:
                m = np.asarray(data[key])
                if m.shape[:2] != tuple(shape[:2]) or not bool(np.all(m[::8, ::8] == 1)):
                    raise NotImplementedError("the device path resizes an all-ones patched_mask of the frame's size")
        data[('image_resize', 'original_shape')] = np.array([shape[0], shape[1]]).astype(int)
        if self.preserve_aspect_ratio:
            scale_factor_x = self.size[0] / shape[0]
            scale_factor_y = self.size[1] / shape[1]
            if self.force_pad:
                scale_factor = min(scale_factor_x, scale_factor_y)
                mode = 'pad_0' if scale_factor_x > scale_factor_y else 'pad_1'
            else:
                scale_factor = scale_factor_x
                mode = 'crop_1' if scale_factor_x > scale_factor_y else 'pad_1'
            h = int(np.round(shape[0] * scale_factor).astype(int))
            w = int(np.round(shape[1] * scale_factor).astype(int))
            scale_factor_yx = (scale_factor, scale_factor)
        else:
            scale_factor_yx = (self.size[0] / shape[0], self.size[1] / shape[1])
            mode, h, w = 'none', self.size[0], self.size[1]
        data[('image_resize', 'effective_size')] = np.array([h, w]).astype(int)
        if len(self.size) <= 1:
            raise NotImplementedError("Resize needs a (height, width) size on the device path")
        plan["resize"] = dict(h=h, w=w, mode=mode, out_h=self.size[0], out_w=self.size[1], keys=list(self.image_keys),
                              gt_keys=list(self.gt_image_keys), src_hw=(shape[0], shape[1]))
        for key in self.calib_keys:
            P = data[key]
            P[0, :] = P[0, :] * scale_factor_yx[1]
            P[1, :] = P[1, :] * scale_factor_yx[0]
            data[key] = P
        return data


class RandomWarpAffine(object):
    """reference :436-497: random scale about a random centre, resized to (output_w, output_h); P2 follows."""
    def __init__(self, scale_lower=0.6, scale_upper=1.4, shift_border=128, output_w=1280, output_h=384,
                 image_keys=['image'], gt_image_keys=[], calib_keys=[], border_mode=0, random_seed=None, **kwargs):
        if border_mode != 0:
            raise NotImplementedError("only cv2.BORDER_CONSTANT (0) is implemented on the device")
        self.scale_lower, self.scale_upper, self.shift_border = scale_lower, scale_upper, shift_border
        self.output_w, self.output_h = output_w, output_h
        self.image_keys, self.gt_image_keys, self.calib_keys = image_keys, gt_image_keys, calib_keys
        self.rng = np.random.default_rng(random_seed if random_seed is not None else np.random.randint(0, 2**32))

    def __call__(self, data):
        height, width = _frame_shape(data, self.image_keys[0])[0:2]
        plan = _plan(data)
        if plan["warp"] is not None or plan["mirror"] or plan["ops"]:
            raise NotImplementedError("the device pipeline warps once, before mirror and colour ops")
        scale = max(height, width) * self.rng.uniform(self.scale_lower, self.scale_upper)
        center_w = self.rng.integers(low=self.shift_border, high=width - self.shift_border)
        center_h = self.rng.integers(low=self.shift_border, high=height - self.shift_border)
        final_scale = max(self.output_w, self.output_h) / scale
        shift_w = self.output_w / 2 - center_w * final_scale
        shift_h = self.output_h / 2 - center_h * final_scale
        plan["warp"] = dict(M=np.array([[final_scale, 0, shift_w], [0, final_scale, shift_h]], dtype=np.float32),
                            out_w=self.output_w, out_h=self.output_h, keys=list(self.image_keys),
                            gt_keys=list(self.gt_image_keys), src_hw=(height, width))
        for key in self.calib_keys:
            P = data[key]
            P[0:2, :] *= final_scale
            P[0, 2] = P[0, 2] + shift_w
            P[0, 3] = P[0, 3] + shift_w * P[2, 3]
            P[1, 2] = P[1, 2] + shift_h
            P[1, 3] = P[1, 3] + shift_h * P[2, 3]
            data[key] = P
        return data


class RandomMirror(object):
    """reference :377-433 (global np.random stream, like the reference)."""
    def __init__(self, mirror_prob, image_keys=['image'], calib_keys=[], gt_image_keys=[], object_keys=[],
                 lidar_keys=[], pose_axis_pairs=[], is_switch_left_right=True, stereo_image_key_pairs=[],
                 stereo_calib_key_pairs=[], **kwargs):
        if object_keys or lidar_keys or stereo_image_key_pairs or stereo_calib_key_pairs:
            raise NotImplementedError("object / lidar / stereo-pair mirroring is outside the monodepth hot path")
        self.mirror_prob = mirror_prob
        self.image_keys, self.calib_keys, self.gt_image_keys = image_keys, calib_keys, gt_image_keys
        self.pose_axis_pairs = pose_axis_pairs

    def __call__(self, data):
        plan = _plan(data)
        if plan["warp"] is not None:
            width = plan["warp"]["out_w"]
        elif plan.get("resize") is not None:
            width = plan["resize"]["out_w"]
        else:
            width = _frame_shape(data, self.image_keys[0])[1]
        if random.rand() <= self.mirror_prob:
            # (the colour ops are per-pixel: a flip before or after them is the same image, so the kernel's fixed
            # order — sample mirrored, then colour — serves configs that mirror first and configs that mirror last)
            plan["mirror"] = not plan["mirror"]
            for key in self.calib_keys:
                P = data[key]
                P[0, 3] = -P[0, 3]
                P[0, 2] = width - P[0, 2] - 1
                data[key] = P
            for key, axis_num in self.pose_axis_pairs:
                data[key] = flip_relative_pose(data[key], axis_num)
        return data


class _ColourOp(object):
    def __init__(self, distort_prob, image_keys, random_seed):
        self.distort_prob, self.image_keys = distort_prob, image_keys
        self.rng = np.random.default_rng(random_seed if random_seed is not None else np.random.randint(0, 2**32))

    def _record(self, data, op, value):
        plan = _plan(data)
        prev = plan.setdefault("colour_keys", list(self.image_keys))
        if list(self.image_keys) != prev:
            raise NotImplementedError("all colour ops of a pipeline must address the same image keys")
        plan["ops"].append((op, value))
        return data


class RandomBrightness(_ColourOp):
    """reference :572-591"""
    def __init__(self, distort_prob, delta=32, image_keys=['image'], random_seed=None, **kwargs):
        assert 0.0 <= delta <= 255.0
        super().__init__(distort_prob, image_keys, random_seed)
        self.delta = delta

    def __call__(self, data):
        value = self.rng.uniform(-self.delta, self.delta) if self.rng.random() <= self.distort_prob else None
        if _plan(data)["hsv"]:
            raise NotImplementedError("brightness is applied in RGB")
        return self._record(data, OP_BRIGHTNESS, value)


class RandomContrast(_ColourOp):
    """reference :545-569"""
    def __init__(self, distort_prob, lower=0.5, upper=1.5, image_keys=['image'], random_seed=None, **kwargs):
        assert upper >= lower >= 0, "contrast bounds"
        super().__init__(distort_prob, image_keys, random_seed)
        self.lower, self.upper = lower, upper

    def __call__(self, data):
        value = self.rng.uniform(self.lower, self.upper) if self.rng.random() <= self.distort_prob else None
        if _plan(data)["hsv"]:
            raise NotImplementedError("contrast is applied in RGB")
        return self._record(data, OP_CONTRAST, value)


class ConvertColor(object):
    """reference :527-542.  RGB->HSV opens a saturation op, HSV->RGB closes it: the kernel fuses the round trip."""
    def __init__(self, current='RGB', transform='HSV', image_keys=['image'], **kwargs):
        if (current, transform) not in (('RGB', 'HSV'), ('HSV', 'RGB')):
            raise NotImplementedError("only RGB<->HSV is implemented on the device")
        self.current, self.transform, self.image_keys = current, transform, image_keys

    def __call__(self, data):
        plan = _plan(data)
        if self.transform == 'HSV':
            if plan["hsv"]:
                raise ValueError("image is already HSV")
            plan["hsv"] = True
            plan["ops"].append((OP_SATURATION, None))       # a bare round trip unless RandomSaturation fills it in
        else:
            if not plan["hsv"]:
                raise ValueError("image is not HSV")
            plan["hsv"] = False
        return data


class RandomSaturation(_ColourOp):
    """reference :200-226 (expects HSV, i.e. between the two ConvertColor stages)."""
    def __init__(self, distort_prob, lower=0.5, upper=1.5, image_keys=['image'], random_seed=None, **kwargs):
        assert upper >= lower >= 0, "saturation bounds"
        super().__init__(distort_prob, image_keys, random_seed)
        self.lower, self.upper = lower, upper

    def __call__(self, data):
        plan = _plan(data)
        value = self.rng.uniform(self.lower, self.upper) if self.rng.random() <= self.distort_prob else None
        if not plan["hsv"] or not plan["ops"] or plan["ops"][-1] != (OP_SATURATION, None):
            raise NotImplementedError("RandomSaturation must directly follow ConvertColor(transform='HSV')")
        plan["ops"][-1] = (OP_SATURATION, value)
        plan.setdefault("colour_keys", list(self.image_keys))
        return data


class Normalize(object):
    """reference :91-109: (x / 255 - mean) / std per channel, recorded per image key."""
    def __init__(self, mean, stds, image_keys=['image'], **kwargs):
        self.mean = np.array(mean, dtype=np.float32)
        self.stds = np.array(stds, dtype=np.float32)
        self.image_keys = image_keys

    def __call__(self, data):
        plan = _plan(data)
        if plan["hsv"]:
            raise ValueError("Normalize on an HSV image")
        for key in self.image_keys:
            if key in plan["normalize"]:
                raise NotImplementedError("one Normalize per image key")
            plan["normalize"][key] = (self.mean.copy(), self.stds.copy())
        return data


class ConvertToTensor(object):
    """reference :62-88.  Calibration / pose entries become tensors; frames stay raw for DeviceAugment."""
    def __init__(self, image_keys=['image'], gt_image_keys=[], calib_keys=[], lidar_keys=[], **kwargs):
        self.image_keys, self.gt_image_keys = image_keys, gt_image_keys
        self.calib_keys, self.lidar_keys = calib_keys, lidar_keys

    def __call__(self, data):
        for key in self.calib_keys + self.lidar_keys:
            data[key] = torch.tensor(data[key], dtype=torch.float32).contiguous()
        return data


# ------------------------------------------------------------------------------------------------
# execution
# ------------------------------------------------------------------------------------------------
def invert_affine(M):
    """cv2.warpAffine's inversion of the forward 2x3 matrix, in float64 (OpenCV imgwarp.cpp)."""
    m = np.asarray(M, dtype=np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = m[4] * D, m[0] * D
    m[0], m[1], m[3], m[4] = a11, m[1] * -D, m[3] * -D, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


class DeviceAugment(object):
    """Batch executor: `collate(samples)` is a DataLoader collate_fn (keeps everything on the host, stacks the raw
    frames into one pinned uint8 buffer), `materialize(batch, device)` runs the kernel; calling the object does both.

    frame_keys: image-key families that share a frame index, default ('image', 'original_image') as in
    mono_dataset.py:188-190 — ('original_image', i) is the unaugmented copy of ('image', i)."""

    def __init__(self, frame_idxs=(0, 1, -1), image_family='image', original_family='original_image',
                 mask_key='patched_mask', device=None):
        self.frame_idxs = list(frame_idxs)
        self.image_family, self.original_family, self.mask_key = image_family, original_family, mask_key
        self.device = device

    # -- host side ---------------------------------------------------------------------------------
    @staticmethod
    def _colour_plan(p, iplan_row, fplan_row):
        """pack one sample's colour ops / mirror flag into the kernels' plan rows (FsAugArgs layout)"""
        ops = list(p["ops"])
        if len(ops) > 3 or len({o for o, _ in ops}) != len(ops):
            raise NotImplementedError("at most one brightness, contrast and saturation op per sample")
        applied = 0
        order = [o for o, _ in ops]
        for o, v in ops:
            if o == OP_SATURATION:
                applied |= 8                      # the HSV round trip itself runs
            if v is not None:
                applied |= 1 << o
                fplan_row[o] = v
        while len(order) < 3:
            order.append(3)                       # 3 = no-op slot
        iplan_row[0:3] = order[:3]
        iplan_row[3] = applied
        iplan_row[4] = int(p["mirror"])

    def _check_normalize(self, s, p, mean, std):
        for idx in self.frame_idxs:
            want = p["normalize"].get((self.image_family, idx), (mean, std))
            if not (np.array_equal(want[0], mean) and np.array_equal(want[1], std)):
                raise NotImplementedError("one mean/std for all augmented frames")
            om, osd = p["normalize"].get((self.original_family, idx), (np.zeros(3, np.float32), np.ones(3, np.float32)))
            if np.any(om != 0) or np.any(osd != 1):
                raise NotImplementedError("('original_image', i) is normalised with mean 0 / std 1")

    def _collate_resize(self, samples, plans):
        B, F = len(samples), len(self.frame_idxs)
        r0 = plans[0]["resize"]
        # training use (Resize-based configs): the resize also feeds ('original_image', i) and the patched mask, and
        # colour ops / a mirror may follow it
        train = any((self.original_family, i) in r0["keys"] for i in self.frame_idxs) or bool(r0.get("gt_keys"))
        if train:
            return self._collate_resize_train(samples, plans)
        Hs = max(p["resize"]["src_hw"][0] for p in plans)
        Ws = max(p["resize"]["src_hw"][1] for p in plans)
        src = torch.zeros(B, F, Hs, Ws, 3, dtype=torch.uint8)
        src_np = src.numpy()
        dims = np.zeros((B, 4), dtype=np.int32)
        key0 = (self.image_family, self.frame_idxs[0])
        mean, std = plans[0]["normalize"].get(key0, (np.zeros(3, np.float32), np.ones(3, np.float32)))
        for b, (s, p) in enumerate(zip(samples, plans)):
            r = p["resize"]
            if p["warp"] is not None or p["ops"] or p["mirror"] or (r["out_h"], r["out_w"]) != (r0["out_h"], r0["out_w"]):
                raise NotImplementedError("a validation batch is Resize + Normalize with one output size")
            h, w = r["src_hw"]
            for f, idx in enumerate(self.frame_idxs):
                src_np[b, f, :h, :w] = s[(self.image_family, idx)]
            # crop_1 keeps the left out_w columns of a wider resize; pads are zero rows / columns after it
            dims[b] = (h, w, r["h"], r["w"])
        batch = {PLAN: dict(src=src, dims=torch.from_numpy(dims), mean=mean, std=std, out_hw=(r0["out_h"], r0["out_w"]),
                            kind="resize")}
        self._collate_rest(samples, batch)
        return batch

    def _collate_resize_train(self, samples, plans):
        B, F = len(samples), len(self.frame_idxs)
        r0 = plans[0]["resize"]
        Hs = max(p["resize"]["src_hw"][0] for p in plans)
        Ws = max(p["resize"]["src_hw"][1] for p in plans)
        src = torch.zeros(B, F, Hs, Ws, 3, dtype=torch.uint8)
        src_np = src.numpy()
        dims = np.zeros((B, 4), dtype=np.int32)
        iplan = np.zeros((B, 8), dtype=np.int32)
        fplan = np.zeros((B, 4), dtype=np.float32)
        key0 = (self.image_family, self.frame_idxs[0])
        mean, std = plans[0]["normalize"].get(key0, (np.zeros(3, np.float32), np.ones(3, np.float32)))
        for b, (s, p) in enumerate(zip(samples, plans)):
            r = p["resize"]
            if p["warp"] is not None or p["hsv"] or (r["out_h"], r["out_w"]) != (r0["out_h"], r0["out_w"]):
                raise NotImplementedError("a Resize-based training batch shares one output size and ends in RGB")
            h, w = r["src_hw"]
            for f, idx in enumerate(self.frame_idxs):
                frame = s[(self.image_family, idx)]
                if frame.shape != (h, w, 3):
                    raise ValueError("frames of one sample must share their size")
                src_np[b, f, :h, :w] = frame
                okey = (self.original_family, idx)
                if okey in s and s[okey] is not frame and (
                        s[okey].shape != frame.shape or not np.array_equal(s[okey][::16, ::16], frame[::16, ::16])):
                    raise ValueError("%r is expected to be the unaugmented copy of the image" % (okey,))
            self._check_normalize(s, p, mean, std)
            dims[b] = (h, w, r["h"], r["w"])
            self._colour_plan(p, iplan[b], fplan[b])
        batch = {PLAN: dict(src=src, dims=torch.from_numpy(dims), iplan=torch.from_numpy(iplan),
                            fplan=torch.from_numpy(fplan), mean=mean, std=std, out_hw=(r0["out_h"], r0["out_w"]),
                            kind="resize", train=True, mask=bool(r0.get("gt_keys")))}
        self._collate_rest(samples, batch)
        return batch

    def _collate_rest(self, samples, batch):
        skip = {PLAN, self.mask_key}
        for idx in self.frame_idxs:
            skip.add((self.image_family, idx)); skip.add((self.original_family, idx))
        for key in samples[0]:
            if key in skip:
                continue
            vals = [s[key] for s in samples]
            if isinstance(vals[0], torch.Tensor):
                batch[key] = torch.stack(vals)
            elif isinstance(vals[0], np.ndarray):
                batch[key] = torch.from_numpy(np.stack(vals))
            else:
                batch[key] = vals

    def collate(self, samples):
        B, F = len(samples), len(self.frame_idxs)
        plans = [s[PLAN] for s in samples]
        if plans[0].get("resize") is not None:
            return self._collate_resize(samples, plans)
        for p in plans:
            if p["warp"] is None:
                raise NotImplementedError("DeviceAugment needs a RandomWarpAffine stage (it fixes the output size)")
            if p["hsv"]:
                raise ValueError("pipeline ended in HSV")
        w0 = plans[0]["warp"]
        out_w, out_h = w0["out_w"], w0["out_h"]
        Hs = max(p["warp"]["src_hw"][0] for p in plans)
        Ws = max(p["warp"]["src_hw"][1] for p in plans)
        ragged = any(p["warp"]["src_hw"] != (Hs, Ws) for p in plans)
        # pageable here (collate may run in a DataLoader worker); the loader's pin_memory thread pins it
        src = (torch.zeros if ragged else torch.empty)(B, F, Hs, Ws, 3, dtype=torch.uint8)
        src_np = src.numpy()
        minv = np.zeros((B, 6), dtype=np.float64)
        iplan = np.zeros((B, 8), dtype=np.int32)
        fplan = np.zeros((B, 4), dtype=np.float32)
        img_key0 = (self.image_family, self.frame_idxs[0])
        mean, std = plans[0]["normalize"].get(img_key0, (np.zeros(3, np.float32), np.ones(3, np.float32)))
        for b, (s, p) in enumerate(zip(samples, plans)):
            h, w = p["warp"]["src_hw"]
            if (p["warp"]["out_w"], p["warp"]["out_h"]) != (out_w, out_h):
                raise ValueError("samples of one batch must share the output size")
            for f, idx in enumerate(self.frame_idxs):
                frame = s[(self.image_family, idx)]
                if frame.shape != (h, w, 3):
                    raise ValueError("frames of one sample must share their size")
                src_np[b, f, :h, :w] = frame
                okey = (self.original_family, idx)
                if okey in s and s[okey] is not frame and (
                        s[okey].shape != frame.shape or not np.array_equal(s[okey][::16, ::16], frame[::16, ::16])):
                    raise ValueError("%r is expected to be the unaugmented copy of the image" % (okey,))
            self._check_normalize(s, p, mean, std)
            minv[b] = invert_affine(p["warp"]["M"])
            self._colour_plan(p, iplan[b], fplan[b])
            iplan[b, 5], iplan[b, 6] = h, w
        batch = {PLAN: dict(src=src, minv=torch.from_numpy(minv), iplan=torch.from_numpy(iplan),
                            fplan=torch.from_numpy(fplan), mean=mean, std=std, out_hw=(out_h, out_w), kind="warp")}
        self._collate_rest(samples, batch)
        return batch

    # -- device side -------------------------------------------------------------------------------
    def materialize(self, batch, device=None):
        from ....hip.binding import lib, check, stream_ptr, FsAugArgs, FsResizeArgs
        device = torch.device(device or self.device or "cuda")
        plan = batch.pop(PLAN)
        src = plan["src"].to(device, non_blocking=True)
        B, F, Hs, Ws, _ = src.shape
        H, W = plan["out_hw"]
        if plan["kind"] == "resize":
            dims = plan["dims"].to(device, non_blocking=True)
            image = torch.empty(F, B, 3, H, W, dtype=torch.float32, device=device)
            r = FsResizeArgs()
            r.src, r.dims, r.image = src.data_ptr(), dims.data_ptr(), image.data_ptr()
            for k in range(3):
                r.mean[k], r.std[k] = float(plan["mean"][k]), float(plan["std"][k])
            r.B, r.F, r.Hs, r.Ws, r.H, r.W = B, F, Hs, Ws, H, W
            original = mask = None
            if plan.get("train"):
                iplan = plan["iplan"].to(device, non_blocking=True)
                fplan = plan["fplan"].to(device, non_blocking=True)
                original = torch.empty(F, B, 3, H, W, dtype=torch.float32, device=device)
                r.iplan, r.fplan, r.original = iplan.data_ptr(), fplan.data_ptr(), original.data_ptr()
                if plan.get("mask"):
                    mask = torch.empty(B, H, W, dtype=torch.float64, device=device)
                    r.mask = mask.data_ptr()
            check(lib.fs_resize_frames(C.byref(r), stream_ptr()), "resize_frames")
            for f, idx in enumerate(self.frame_idxs):
                batch[(self.image_family, idx)] = image[f]
                if original is not None:
                    batch[(self.original_family, idx)] = original[f]
            if mask is not None:
                batch[self.mask_key] = mask
            for key, val in list(batch.items()):
                if isinstance(val, torch.Tensor) and not val.is_cuda:
                    batch[key] = val.to(device, non_blocking=True)
            return batch
        minv = plan["minv"].to(device, non_blocking=True)
        iplan = plan["iplan"].to(device, non_blocking=True)
        fplan = plan["fplan"].to(device, non_blocking=True)
        image = torch.empty(F, B, 3, H, W, dtype=torch.float32, device=device)
        original = torch.empty(F, B, 3, H, W, dtype=torch.float32, device=device)
        mask = torch.empty(B, H, W, dtype=torch.float64, device=device)
        a = FsAugArgs()
        a.src, a.minv, a.iplan, a.fplan = src.data_ptr(), minv.data_ptr(), iplan.data_ptr(), fplan.data_ptr()
        a.image, a.original, a.mask = image.data_ptr(), original.data_ptr(), mask.data_ptr()
        for k in range(3):
            a.mean[k], a.std[k] = float(plan["mean"][k]), float(plan["std"][k])
        a.B, a.F, a.Hs, a.Ws, a.H, a.W = B, F, Hs, Ws, H, W
        check(lib.fs_augment_frames(C.byref(a), stream_ptr()), "augment_frames")
        for f, idx in enumerate(self.frame_idxs):
            batch[(self.image_family, idx)] = image[f]
            batch[(self.original_family, idx)] = original[f]
        batch[self.mask_key] = mask
        for key, val in list(batch.items()):
            if isinstance(val, torch.Tensor) and not val.is_cuda:
                batch[key] = val.to(device, non_blocking=True)
        return batch

    def __call__(self, samples, device=None):
        return self.materialize(self.collate(samples), device)
