"""Time fs_photo_loss_bwd (and the other loss kernels) alone at the bench shape.
python tools/probes/photo_bwd_time.py [B H W]   (FSNET_HIP_LIB selects a library variant)"""
import ctypes as C
import sys
import torch
from fsnet_amd.hip import ops
from fsnet_amd.hip.binding import lib, check, stream_ptr

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (12, 192, 640)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
base = torch.rand(B, 3, H // 8, W // 8, generator=g)
img = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear").contiguous().to(dev)
srcs = [torch.roll(img, 3, 3).contiguous() + 0.02 * torch.rand(B, 3, H, W, generator=g).to(dev),
        torch.roll(img, -3, 3).contiguous() + 0.02 * torch.rand(B, 3, H, W, generator=g).to(dev)]
P2 = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0]]).repeat(B, 1, 1).to(dev)
T0 = torch.eye(4).repeat(B, 1, 1); T0[:, 0, 3] = 0.3; T0[:, 2, 3] = -0.5
T1 = torch.eye(4).repeat(B, 1, 1); T1[:, 0, 3] = -0.3; T1[:, 2, 3] = 0.5
Ts = [T0.to(dev), T1.to(dev)]
S = 4
depths = [(torch.rand(B, 1, H >> s, W >> s, generator=g) * 20 + 5).to(dev) for s in range(S)]
disps = [1.0 / d for d in depths]
pl = ops.PhotometricLoss(B, H, W, [0, 1, 2, 3], dev, 0.5, 100.0)
out = pl.forward(img, srcs, P2, Ts, None, depths, disps, noise_seed=1)
pl.backward(None)
torch.cuda.synchronize()
sel = pl.sel.view(S, B, H, W)
print("selection histogram:", [int((sel == k).sum()) for k in range(4)])
pa = C.byref(pl._pa)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st = stream_ptr()
print("photo_loss_bwd  %.1f us" % timeit(lambda: check(lib.fs_photo_loss_bwd(pa, st), "b")))
print("photo_loss_fwd  %.1f us" % timeit(lambda: check(lib.fs_photo_loss_fwd(pa, st), "f")))
print("photo_warp      %.1f us" % timeit(lambda: check(lib.fs_photo_warp(pa, st), "w")))
print("photo_ident     %.1f us" % timeit(lambda: check(lib.fs_photo_identity(pa, st), "i")))
