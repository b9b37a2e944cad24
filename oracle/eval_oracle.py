"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's depth evaluation, SURVEY 8f rank 2.
Only tests/ may import this module; the product path is fsnet_amd/csrc/eval.hip.

  compute_errors      monodepth/networks/utils/monodepth_utils.py:271-289      PINNED: tests/golden/eval.npz holds inputs
                                                                               and the outputs of the real function
  single_loss         monodepth/evaluation/kitti_unsupervised_eval.py:47-80    restated line by line
  cv2_resize_linear   OpenCV (opencv-python, unpinned in the reference's requirements) imgproc/resize.cpp,
                      INTER_LINEAR, single-channel CV_32F.  cv2 is absent from this image and from
                      /root/reference: PARITY UNPINNED for this function (restated from the published algorithm:
                      sample position (d + 0.5) * scale - 0.5, floor, fraction zeroed and index clamped at both
                      borders, horizontal then vertical pass in float32).
"""
import numpy as np


def compute_errors(gt, pred):
    thresh = np.maximum((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def _lin_coords(n_dst, n_src):
    scale = np.float32(n_src) / np.float32(n_dst)
    f = (np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= n_src - 1
    f[hi] = 0; s[hi] = n_src - 1
    return s, np.minimum(s + 1, n_src - 1), f


def cv2_resize_linear(src, W, H):
    """cv2.resize(src, (W, H)) for a float32 single-channel image (default interpolation INTER_LINEAR)."""
    src = np.asarray(src, dtype=np.float32)
    h, w = src.shape
    if (h, w) == (H, W):
        return src.copy()
    x0, x1, fx = _lin_coords(W, w)
    y0, y1, fy = _lin_coords(H, h)
    one = np.float32(1)
    top = src[y0][:, x0] * (one - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (one - fx) + src[y1][:, x1] * fx
    return (top * (one - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


def single_loss(depth_0, gt_depth):
    gt_height, gt_width = gt_depth.shape[:2]
    pred_depth = cv2_resize_linear(depth_0, gt_width, gt_height)
    mask = np.logical_and(gt_depth > 1e-3, gt_depth < 80.0)
    crop = np.array([0.40810811 * gt_height, 0.99189189 * gt_height,
                     0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)
    crop_mask = np.zeros(mask.shape)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
    mask = np.logical_and(mask, crop_mask)
    pred_depth = pred_depth[mask]
    gt_depth = gt_depth[mask]
    if len(pred_depth) == 0 or len(gt_depth) == 0:
        raise ValueError
    ratio = np.median(gt_depth) / np.median(pred_depth)
    scaled_depth = pred_depth * ratio
    scaled_depth[scaled_depth < 1e-3] = 1e-3
    scaled_depth[scaled_depth > 80.0] = 80.0
    error = compute_errors(gt_depth, scaled_depth)
    pred_depth[pred_depth < 1e-3] = 1e-3
    pred_depth[pred_depth > 80.0] = 80.0
    abs_error = compute_errors(gt_depth, pred_depth)
    return dict(ratio=ratio, error=error, abs_error=abs_error)
