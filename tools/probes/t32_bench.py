"""Stand-alone timing of the 3x3/s1 convolution kernels on the step's layer shapes: the 16x16-tile kernel
(FSNET_AMD_T32=0) against every tile configuration of the 32x32-tile kernel (FSNET_AMD_T32_CFG).  HIP-event time over
back-to-back launches.  Usage: python tools/probes/t32_bench.py [reps]"""
import os
import sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda:0'); dt = torch.bfloat16
SHAPES = [(64, 64, 48, 160, 12), (64, 64, 48, 160, 36), (128, 128, 24, 80, 12), (128, 128, 24, 80, 36),
          (256, 256, 12, 40, 12), (256, 256, 12, 40, 36), (512, 512, 6, 20, 12), (512, 512, 6, 20, 36),
          (64, 64, 80, 256, 8), (128, 64, 48, 160, 12), (512, 256, 12, 40, 12)]


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def setcfg(c):
    if c == "old":
        os.environ["FSNET_AMD_T32"] = "0"
        os.environ.pop("FSNET_AMD_T32_CFG", None)
    else:
        os.environ["FSNET_AMD_T32"] = "1"
        if c == "auto":
            os.environ.pop("FSNET_AMD_T32_CFG", None)
        else:
            os.environ["FSNET_AMD_T32_CFG"] = str(c)


print("%-28s %-6s %9s %9s %9s   (us; TFLOP/s fwd)" % ("shape", "cfg", "fwd+stat", "dgrad", "dgrad+bnb"))
for Ci, Co, H, W, B in SHAPES:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
    gy = torch.randn_like(y)
    dx = torch.empty_like(x)
    mask = torch.randn_like(x)
    cin = torch.randn_like(x)
    st = type("S", (), {})()
    st.mean = torch.randn(op.Ci_p, device=dev); st.invstd = torch.rand(op.Ci_p, device=dev) + 0.5; st.groups = 1
    sums = torch.zeros(8, 2, op.Ci_p, dtype=torch.float64, device=dev)
    flops = 2.0 * B * H * W * Co * 9 * Ci
    for cfg in ["old", 0, 1, 2, 3, "auto"]:
        setcfg(cfg)
        tf = timed(lambda: op.forward(x, out=y, stats=stats))
        td = timed(lambda: op.dgrad(gy, H, W, out=dx))
        tb = timed(lambda: op.dgrad(gy, H, W, out=dx, mask=mask, bn_fuse=(cin, st, sums)))
        print("%-28s %-6s %9.1f %9.1f %9.1f   %6.0f" % ("%d->%d @%dx%d B=%d" % (Ci, Co, H, W, B), cfg, tf, td, tb,
                                                     flops / tf * 1e-6), flush=True)
