"""Small helpers with the reference's names (monodepth/networks/utils/monodepth_utils.py:8-63).
disp/depth conversions are trivial scalar formulas kept for API compatibility; the fused training
path computes them inside the HIP head kernels."""
import torch

from fsnet_amd.hip import ops


def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


def depth_to_disp(depth, min_depth, max_depth):
    return (1 / depth - 1 / max_depth) / (1 / min_depth - 1 / max_depth)


class _TransformFn(torch.autograd.Function):
    """(axisangle [B,1,3], translation [B,1,3]) -> 4x4 via the pose-tail kernel (hw=1, scale=1)."""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        B = axisangle.shape[0]
        x = torch.zeros(B, 1, 1, 16, dtype=torch.float32, device=axisangle.device)
        x[:, 0, 0, 0:3] = axisangle.reshape(B, 3)
        x[:, 0, 0, 3:6] = translation.reshape(B, 3)
        _, _, T = ops.pose_tail_fwd(x, 1, invert, scale=1.0)
        ctx.save_for_backward(x)
        ctx.invert = invert
        return T

    @staticmethod
    def backward(ctx, gT):
        (x,) = ctx.saved_tensors
        d = ops.pose_tail_bwd(x, gT.contiguous().float(), 1, ctx.invert, torch.float32, scale=1.0)
        B = x.shape[0]
        return d[:, 0, 0, 0:3].reshape(B, 1, 3), d[:, 0, 0, 3:6].reshape(B, 1, 3), None


def transformation_from_parameters(axisangle, translation, invert=False):
    """Compatibility entry (monodepth_utils.py:45-63); the training path uses the fused pose tail."""
    return _TransformFn.apply(axisangle, translation, bool(invert))


# ----------------------------------------------------------------------------------------------------------------
# KITTI ground-truth depth from raw velodyne scans (reference monodepth_utils.py:291-295, 339-420 — the monodepth2 /
# KITTI-matlab export).  Host-side numpy, run once per split by KittiEigenEvaluator; results are cached in an .npz.
# ----------------------------------------------------------------------------------------------------------------
def read_calib_file(path):
    """KITTI calibration text file -> {key: float array | string}"""
    import numpy as np
    numeric = set("0123456789.e+- ")
    out = {}
    with open(path, "r") as f:
        for line in f:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            value = value.strip()
            out[key] = value
            if numeric.issuperset(value):
                try:
                    out[key] = np.array([float(v) for v in value.split(" ")])
                except ValueError:
                    pass
    return out


def load_velodyne_points(filename):
    """[N, 4] float32 (forward, left, up, reflectance) with the last column set to 1 (homogeneous)"""
    import numpy as np
    pts = np.fromfile(filename, dtype=np.float32).reshape(-1, 4)
    pts[:, 3] = 1.0
    return pts


def sub2ind(matrix_size, row, col):
    """the export's linear index, kept EXACTLY as the reference has it (row * (n - 1) + col - 1: not a bijection —
    distinct pixels can share an index, and the duplicate pass below then mixes them; ground truth made with the
    reference has that property, so an evaluator that wants the same numbers must have it too)"""
    m, n = matrix_size
    return row * (n - 1) + col - 1


def generate_depth_map(calib_dir, velo_filename, cam=2, vel_depth=False):
    """velodyne scan -> sparse depth image of camera `cam` ([H, W] float64, 0 = no return)"""
    import os
    import numpy as np
    cam2cam = read_calib_file(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    v2c = read_calib_file(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
    velo2cam = np.vstack((np.hstack((v2c["R"].reshape(3, 3), v2c["T"][..., np.newaxis])), np.array([0, 0, 0, 1.0])))
    im_shape = cam2cam["S_rect_02"][::-1].astype(np.int32)
    R_cam2rect = np.eye(4)
    R_cam2rect[:3, :3] = cam2cam["R_rect_00"].reshape(3, 3)
    P_velo2im = np.dot(np.dot(cam2cam["P_rect_0" + str(cam)].reshape(3, 4), R_cam2rect), velo2cam)

    velo = load_velodyne_points(velo_filename)
    velo = velo[velo[:, 0] >= 0, :]                       # in front of the sensor (approximation of "in front of the camera")
    pts = np.dot(P_velo2im, velo.T).T
    pts[:, :2] = pts[:, :2] / pts[:, 2][..., np.newaxis]
    if vel_depth:
        pts[:, 2] = velo[:, 0]
    pts[:, 0] = np.round(pts[:, 0]) - 1                   # (- 1: the KITTI matlab convention)
    pts[:, 1] = np.round(pts[:, 1]) - 1
    ok = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < im_shape[1]) & (pts[:, 1] < im_shape[0])
    pts = pts[ok, :]
    rows, cols = pts[:, 1].astype(int), pts[:, 0].astype(int)
    depth = np.zeros((im_shape[:2]))
    depth[rows, cols] = pts[:, 2]                         # (last writer wins, as in the reference's fancy assignment)
    # points sharing an export index: the pixel of the FIRST of them gets the smallest depth of the group
    inds = sub2ind(depth.shape, pts[:, 1], pts[:, 0])
    _, first, inverse, counts = np.unique(inds, return_index=True, return_inverse=True, return_counts=True)
    gmin = np.full(len(first), np.inf)
    np.minimum.at(gmin, inverse, pts[:, 2])
    dupe = counts > 1
    depth[rows[first[dupe]], cols[first[dupe]]] = gmin[dupe]
    depth[depth < 0] = 0
    return depth
