"""stand-alone time of the BatchNorm passes on encoder shapes (HIP events, median of 30): are they slow alone, or only in the
step?  bytes = tensors read + written"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fsnet_amd.hip import ops
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


SHAPES = [(12, 48, 160, 64), (24, 48, 160, 64), (12, 24, 80, 128), (12, 12, 40, 256), (12, 6, 20, 512), (12, 192, 640, 16)]
if len(sys.argv) > 1 and sys.argv[1] == "r50":       # ResNet-50 at 320x1024: the wide tensors of layers 1-4, pose (16) and depth (8) batch
    SHAPES = [(16, 80, 256, 256), (8, 80, 256, 256), (16, 40, 128, 512), (16, 20, 64, 1024), (16, 10, 32, 2048), (16, 80, 256, 64)]
for N, H, W, C in SHAPES + [(12, 48, 160, 64), (24, 48, 160, 64), (12, 24, 80, 128), (12, 12, 40, 256), (12, 6, 20, 512), (12, 192, 640, 16)][:0]:
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    g = torch.randn(N, H, W, C, device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    stats = torch.rand(8, 2, C, dtype=torch.float64, device=dev) * 100 + 1000
    bn = {"weight": torch.ones(C, device=dev), "bias": torch.zeros(C, device=dev), "running_mean": torch.zeros(C, device=dev),
          "running_var": torch.ones(C, device=dev), "num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
    st = ops.BnState(C, dev)
    mb = x.numel() * 2 / 1e6
    t_apply = tm(lambda: ops.bn_apply(x, stats, bn, st, y, H, W, N * H * W, relu=True))
    sums = torch.rand(8, 2, C, dtype=torch.float64, device=dev)
    t_bwd = tm(lambda: ops.bn_backward(g, None, x, bn["weight"], st, dx, None, None, H, W, sums=sums, reduced=True))
    yy = torch.randn(N, H, W, C, device=dev).bfloat16()
    t_red = tm(lambda: ops.bn_backward(g, yy, x, bn["weight"], st, dx, None, None, H, W, sums=sums, sums_zeroed=True, phase="reduce"))
    t_copy = tm(lambda: y.copy_(x))
    print("[%2d,%3d,%3d,%3d] %5.1f MB/tensor  bn_apply %5.1f us (%.2f TB/s)  bn_bwd_apply %5.1f us (%.2f TB/s)  bn_bwd_reduce %5.1f us (%.2f TB/s)  copy %5.1f us (%.2f TB/s)" % (
        N, H, W, C, mb, t_apply, 2 * mb / t_apply, t_bwd, 3 * mb / t_bwd, t_red, 3 * mb / t_red, t_copy, 2 * mb / t_copy))
