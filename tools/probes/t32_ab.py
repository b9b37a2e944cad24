"""same-process A/B of conv kernel variants: rounds interleave the variants (median over rounds).
variants: old (16x16-tile kernel) | 1 | 2 | 3 (32x32-tile kernel, tile configuration)"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
VARS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["old", "3", "1", "2"]
SHAPES = [(64, 64, 48, 160, 12), (64, 64, 48, 160, 36), (128, 128, 24, 80, 36), (256, 256, 12, 40, 36), (512, 512, 6, 20, 36)]
reps, rounds = 20, 5
def setv(v):
    if v == "old":
        os.environ["FSNET_AMD_T32"] = "0"
    else:
        os.environ["FSNET_AMD_T32"] = "1"; os.environ["FSNET_AMD_T32_CFG"] = v
def timed(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
print("%-26s " % "shape" + " ".join("%8s" % v for v in VARS) + "   (fwd+stats us, median of %d rounds; then dgrad+bnb)" % rounds)
for Ci, Co, H, W, B in SHAPES:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
    gy = torch.randn_like(y); dx = torch.empty_like(x); mask = torch.randn_like(x); cin = torch.randn_like(x)
    st = type("S", (), {})()
    st.mean = torch.randn(op.Ci_p, device=dev); st.invstd = torch.rand(op.Ci_p, device=dev) + 0.5; st.groups = 1
    sums = torch.zeros(8, 2, op.Ci_p, dtype=torch.float64, device=dev)
    for name, fn in [("fwd", lambda: op.forward(x, out=y, stats=stats)),
                     ("dgb", lambda: op.dgrad(gy, H, W, out=dx, mask=mask, bn_fuse=(cin, st, sums)))]:
        res = {v: [] for v in VARS}
        for r in range(rounds + 1):
            for v in VARS:
                setv(v)
                tt = timed(fn)
                if r > 0: res[v].append(tt)
        print("%-26s " % ("%d->%d@%dx%d B%d %s" % (Ci, Co, H, W, B, name)) +
              " ".join("%8.1f" % sorted(res[v])[len(res[v]) // 2] for v in VARS), flush=True)
