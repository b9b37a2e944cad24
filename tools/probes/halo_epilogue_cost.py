"""What the fused BatchNorm statistics cost in the halo conv's epilogue: forward launches with and without `stats`,
timed with HIP events (kernel alone on the chip).  python tools/probes/halo_epilogue_cost.py"""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
for (C, H, W, B) in [(64, 48, 160, 12), (64, 48, 160, 24), (128, 24, 80, 12), (256, 12, 40, 12), (512, 6, 20, 12)]:
    op = ConvOp(C, C, 3, 3, 1, 1, dt, dev)
    op.pack(torch.randn(C, C, 3, 3, device=dev) * 0.05)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(x, stats=stats)
    res = []
    for st in (stats, None):
        for _ in range(10):
            op.forward(x, out=y, stats=st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200):
            op.forward(x, out=y, stats=st)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 200 * 1e3)
    fl = 2.0 * B * H * W * C * C * 9
    print("C=%d %dx%d B=%d: with stats %.1f us (%.0f TF), without %.1f us (%.0f TF)" % (
        C, H, W, B, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6))
