"""3x3 weight gradient (kernel + reduce) under sets of environment variables: values against the first set, time of each.
args: one set per argument, NAME=V[,NAME=V] (`none` = nothing set).  The shipped library reads no development switch;
this is the harness the round-3 experiments (tap groups, block counts, atomic epilogue) were timed with."""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
MODES = sys.argv[1:]
SET = []
dev = torch.device('cuda:0'); dt = torch.bfloat16
shapes = [(64, 64, 48, 160, 12), (64, 64, 48, 160, 24), (128, 128, 24, 80, 12), (128, 128, 24, 80, 24), (256, 256, 12, 40, 24),
          (512, 512, 6, 20, 24), (64, 64, 96, 320, 12), (128, 64, 24, 80, 12)]
for Ci, Co, H, W, B in shapes:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    gy = torch.randn(B, H, W, op.Co_p, device=dev).to(dt)
    res = {}
    for mode in MODES:
        for k in SET:
            os.environ.pop(k, None)
        SET.clear()
        for kv in mode.split(","):
            if "=" in kv:
                os.environ[kv.split("=")[0]] = kv.split("=")[1]
                SET.append(kv.split("=")[0])
        dw = torch.zeros(Co, Ci, 3, 3, device=dev)
        op.wgrad(gy, x, dw)
        torch.cuda.synchronize()
        ref = dw.clone()
        for _ in range(5):
            op.wgrad(gy, x, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            op.wgrad(gy, x, dw)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (ref, e0.elapsed_time(e1) / 50 * 1e3)
    ref = res[MODES[0]][0]
    print("%-24s" % ((Ci, Co, H, W, B),), "  ".join("%s %.1f us (d %.2g)" % (m, res[m][1], (res[m][0] - ref).abs().max().item() / ref.abs().max().item()) for m in MODES))
