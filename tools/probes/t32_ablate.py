"""ablation of the 32x32-tile conv kernel on one shape: which part of a launch costs what (FSNET_AMD_T32_ABL bits:
1 no MFMA loop, 2 no epilogue, 4 no activation loads, 8 no weight loads)"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
reps = 30
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for Ci, Co, H, W, B in [(64, 64, 48, 160, 12), (64, 64, 48, 160, 36), (256, 256, 12, 40, 36)]:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
    for cfg in [3, 1, 0]:
        os.environ["FSNET_AMD_T32_CFG"] = str(cfg)
        row = []
        for abl in [0, 1, 2, 3, 4, 8, 12, 13, 15]:
            os.environ["FSNET_AMD_T32_ABL"] = str(abl)
            row.append("%d:%5.1f" % (abl, timed(lambda: op.forward(x, out=y, stats=stats))))
        print("%d->%d @%dx%d B=%d cfg %d | " % (Ci, Co, H, W, B, cfg) + "  ".join(row), flush=True)
