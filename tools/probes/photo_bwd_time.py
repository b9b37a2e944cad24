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
_dP_big = torch.zeros(S * B * int(lib.fs_photo_bwd_tiles(H, W)), 2, 12, device=dev)   # the staged kernel's tile count
pl._pa.dP = _dP_big.data_ptr()
print("photo_loss_bwd  %.1f us" % timeit(lambda: check(lib.fs_photo_loss_bwd(pa, st), "b")))
print("photo_loss_fwd  %.1f us" % timeit(lambda: check(lib.fs_photo_loss_fwd(pa, st), "f")))
print("photo_warp      %.1f us" % timeit(lambda: check(lib.fs_photo_warp(pa, st), "w")))
print("photo_ident     %.1f us" % timeit(lambda: check(lib.fs_photo_identity(pa, st), "i")))
print("photo_fused_fwd (pred written) %.1f us" % timeit(lambda: check(lib.fs_photo_fused_fwd(pa, st), "ff")))
pl.loss_sums.zero_(); check(lib.fs_photo_warp(pa, st), "w"); check(lib.fs_photo_loss_fwd(pa, st), "f"); torch.cuda.synchronize()
ls_staged, sel_staged = pl.loss_sums.clone(), pl.sel.clone()
pl.loss_sums.zero_(); check(lib.fs_photo_fused_fwd(pa, st), "ff"); torch.cuda.synchronize()
print("fused vs staged: loss sums max rel dev %.3e, selection mismatches %d of %d" % (
    float(((pl.loss_sums - ls_staged).abs() / ls_staged.abs().clamp_min(1e-12)).max()), int((pl.sel != sel_staged).sum()), pl.sel.numel()))
# backward: staged kernel (reads pred) against the fused one, same sel / pred
def run_bwd(fused):
    pl._dd_flat.zero_()
    tiles = int(lib.fs_photo_fused_bwd_tiles(H, W) if fused else lib.fs_photo_bwd_tiles(H, W))
    dP = torch.zeros(S * B * tiles, 2, 12, device=dev)
    pl._pa.dP = dP.data_ptr()
    check((lib.fs_photo_fused_bwd if fused else lib.fs_photo_loss_bwd)(pa, st), "bwd")
    dT = [torch.zeros(B, 4, 4, device=dev) for _ in range(2)]
    check(lib.fs_photo_pose_grad(pl.geo.data_ptr(), dP.data_ptr(), dT[0].data_ptr(), dT[1].data_ptr(), B, S, tiles, st), "pg")
    torch.cuda.synchronize()
    return [d.clone() for d in pl.d_depth], dT, dP
dd_s, dT_s, _ = run_bwd(False)
dd_f, dT_f, dPf = run_bwd(True)
for k in range(S):
    print("  d_depth[%d]: rel L2 dev %.3e (norm %.3e)" % (k, float((dd_f[k] - dd_s[k]).norm() / dd_s[k].norm()), float(dd_s[k].norm())))
for k in range(2):
    print("  dT[%d]: max abs dev %.3e (max %.3e)" % (k, float((dT_f[k] - dT_s[k]).abs().max()), float(dT_s[k].abs().max())))
print("photo_fused_bwd                %.1f us" % timeit(lambda: check(lib.fs_photo_fused_bwd(pa, st), "fb")))
pl._pa.pred, pl._pa.ov = None, None
print("photo_fused_fwd (no pred)      %.1f us" % timeit(lambda: check(lib.fs_photo_fused_fwd(pa, st), "ff")))
