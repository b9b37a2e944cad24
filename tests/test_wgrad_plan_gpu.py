"""The split-K weight-gradient grids must fit ONE round of resident blocks: 288 blocks of a kernel that holds one block
per CU ran as two rounds and took twice as long.  fs_conv_wgrad_plan reports the launch the library
would make; the shapes are the step's (ResNet-18 stages at 192x640, batch 12 and the stacked pose pass, the decoder's
16/32-channel layers, the 7x7 stems)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # Ci, Co, k, stride, pad, H, W, B
    (64, 64, 3, 1, 1, 48, 160, 12), (64, 64, 3, 1, 1, 48, 160, 24), (128, 128, 3, 1, 1, 24, 80, 12),
    (128, 128, 3, 1, 1, 24, 80, 24), (256, 256, 3, 1, 1, 12, 40, 24), (512, 512, 3, 1, 1, 6, 20, 12),
    (512, 512, 3, 1, 1, 6, 20, 24), (64, 64, 3, 1, 1, 80, 256, 8), (16, 16, 3, 1, 1, 192, 640, 12),
    (96, 32, 3, 1, 1, 96, 320, 12), (32, 16, 3, 1, 1, 96, 320, 12), (3, 64, 7, 2, 3, 192, 640, 12),
    (6, 64, 7, 2, 3, 192, 640, 24),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_split_k_grid_is_one_round(dev, shape):
    from fsnet_amd.hip.binding import FsWgradArgs, lib, FS_DTYPE_BF16
    from fsnet_amd.hip.conv import ConvOp, wgrad_workspace, _nhwc_strides, _span_bytes
    Ci, Co, k, stride, pad, H, W, B = shape
    dt = torch.bfloat16
    op = ConvOp(Ci, Co, k, k, stride, pad, dt, dev)
    Ho, Wo = op.out_hw(H, W)
    x = torch.zeros(B, H, W, op.Ci_p, dtype=dt, device=dev)
    dy = torch.zeros(B, Ho, Wo, op.Co_p, dtype=dt, device=dev)
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    a = FsWgradArgs()
    a.dy, a.x, a.dw, a.ktab = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), op.ktab_w.data_ptr()
    a.sN, a.sH, a.sW = _nhwc_strides(x)
    a.Hs, a.Ws, a.Hd, a.Wd = H, W, Ho, Wo
    a.M, a.Cd = B * Ho * Wo, op.Co_p
    a.Co, a.Ci, a.R, a.S = Co, Ci, k, k
    a.stride, a.pad, a.ncolgroups = stride, pad, op.ncolgroups
    ws = wgrad_workspace(dev)
    a.workspace, a.workspace_elems = ws.data_ptr(), ws.numel()
    a.x_bytes, a.use_halo = _span_bytes(x), 1
    plan = (C.c_int32 * 4)()
    assert lib.fs_conv_wgrad_plan(C.byref(a), FS_DTYPE_BF16, plan) == 0
    kind, blocks, threads, resident = list(plan)
    assert blocks >= 1 and threads in (256, 512) and resident >= 128
    if kind in (1, 2, 3):          # the LDS-halo family: blocks are long latency chains, a second round doubles the kernel
        assert blocks <= resident, (shape, list(plan))
    else:                          # generic tile kernel: short blocks, sized to stay within one round as well
        assert blocks <= resident, (shape, list(plan))
    # and the launch itself still works on these arguments
    assert lib.fs_conv_wgrad(C.byref(a), FS_DTYPE_BF16, None) == 0
    torch.cuda.synchronize()
