"""stand-alone time of the decoder's one-chunk 3x3 layers: persistent kernel (conv3x3_p1.hip) against the LDS-halo kernel
(FSNET_AMD_P1=0 in a second process); HIP events, median of 30; bytes = input + output"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fsnet_amd.hip.conv import ConvOp
dev = torch.device("cuda:0")


def tm(fn, n=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for Ci, Co, N, H, W, f32out in [(16, 16, 12, 192, 640, False), (16, 16, 12, 192, 640, True), (32, 16, 12, 96, 320, True), (32, 16, 12, 96, 320, False)]:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, torch.bfloat16, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) / 12)
    x = torch.randn(N, H, W, Ci, device=dev).bfloat16()
    gy = torch.randn(N, H, W, op.Co_p, device=dev).bfloat16()
    bias = torch.zeros(op.Co_p, device=dev)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    out = torch.empty(N, H, W, op.Co_p, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    dx = torch.empty(N, H, W, op.Ci_p, device=dev, dtype=torch.bfloat16)
    tf = tm(lambda: op.forward(x, out=out, bias=bias, stats=None if f32out else stats, out_f32=f32out))
    td = tm(lambda: op.dgrad(gy, H, W, out=dx))
    mbf = (x.numel() * 2 + out.numel() * out.element_size()) / 1e6
    mbd = (gy.numel() * 2 + dx.numel() * 2) / 1e6
    print("%2d->%2d @%dx%d %s  %s  fwd %5.1f us (%.2f TB/s)   dgrad %5.1f us (%.2f TB/s)" % (
        Ci, Co, H, W, "f32out" if f32out else "bf16  ", op.plan_3x3(N, H, W)["kernel"], tf, mbf / tf, td, mbd / td))
