"""stand-alone time of the stride-2 3x3 forward convolutions of the ResNet stage entries: LDS-halo stride-2 variant
(FSNET_AMD_S2_CO = 16 / 32) against the implicit GEMM; HIP events, median of 30"""
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


for Ci, Co, N, H, W in [(64, 128, 12, 48, 160), (64, 128, 24, 48, 160), (128, 256, 12, 24, 80), (128, 256, 24, 24, 80),
                        (256, 512, 12, 12, 40), (256, 512, 24, 12, 40)]:
    op = ConvOp(Ci, Co, 3, 3, 2, 1, torch.bfloat16, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) / 30)
    x = torch.randn(N, H, W, Ci, device=dev).bfloat16()
    stats = torch.zeros(8, 2, Co, dtype=torch.float64, device=dev)
    out = torch.empty(N, H // 2, W // 2, Co, device=dev, dtype=torch.bfloat16)
    res = {}
    for name, env, s2 in (("igemm", None, False), ("halo16", "16", True), ("halo32", "32", True)):
        op.halo_f_s2 = s2
        if env:
            os.environ["FSNET_AMD_S2_CO"] = env
        res[name] = tm(lambda: op.forward(x, out=out, stats=stats))
        os.environ.pop("FSNET_AMD_S2_CO", None)
    gf = 2.0 * N * (H // 2) * (W // 2) * Co * 9 * Ci / 1e9
    print("%3d->%3d N=%2d %3dx%3d  %.2f GF   " % (Ci, Co, N, H, W, gf) + "  ".join("%s %.1f us" % kv for kv in res.items()))
