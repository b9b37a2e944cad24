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
