"""GPU parity of the non-conv HIP kernels against the CPU oracle (oracle/fsnet_oracle.py) and the
golden vectors from the real reference.  All calls go through the C ABI (ctypes)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fsnet_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def nhwc(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw(x):
    return x.permute(0, 3, 1, 2).float().cpu()


def bn_dict(C, g, dev):
    return {"weight": (1 + 0.2 * torch.randn(C, generator=g)).to(dev), "bias": (0.1 * torch.randn(C, generator=g)).to(dev),
            "running_mean": torch.zeros(C, device=dev), "running_var": torch.ones(C, device=dev),
            "num_batches_tracked": torch.zeros((), dtype=torch.long, device=dev)}


def stats_of(x):  # x NCHW cpu -> [FS_STAT_SLOTS][2][C] with everything in slot 3
    d = x.double()
    out = torch.zeros(8, 2, x.shape[1], dtype=torch.float64)
    out[3] = torch.stack([d.sum(dim=(0, 2, 3)), (d * d).sum(dim=(0, 2, 3))])
    return out


@pytest.mark.parametrize("mode", ["plain", "res", "res_bn2", "pad_fold"])
def test_bn_forward_backward(dev, mode):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 3, 32, 10, 14
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    bn = bn_dict(C, g, dev)
    xr = x.clone().requires_grad_(True)
    gam = bn["weight"].cpu().clone().requires_grad_(True)
    bet = bn["bias"].cpu().clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    out = F.batch_norm(xr, rm, rv, gam, bet, training=True, momentum=0.1, eps=1e-5)
    res = res_r = None
    bn2 = st2 = None
    if mode in ("res", "res_bn2"):
        res = torch.randn(N, C, H, W, generator=g)
        res_r = res.clone().requires_grad_(True)
        if mode == "res_bn2":
            bn2 = bn_dict(C, g, dev)
            g2 = bn2["weight"].cpu().clone().requires_grad_(True)
            b2 = bn2["bias"].cpu().clone().requires_grad_(True)
            out = out + F.batch_norm(res_r, torch.zeros(C), torch.ones(C), g2, b2, training=True, eps=1e-5)
        else:
            out = out + res_r
    y_ref = F.relu(out)
    gy = torch.randn(N, C, H, W, generator=g)
    gpad = None
    if mode == "pad_fold":
        # consumer saw the replicate-padded tensor: its gradient lives on the padded grid
        yp = F.pad(y_ref, (1, 1, 1, 1), mode="replicate")
        gpad = torch.randn(N, C, H + 2, W + 2, generator=g)
        yp.backward(gpad)
    else:
        y_ref.backward(gy)

    xd = nhwc(x).to(dev)
    stats = stats_of(x).to(dev)
    st = ops.BnState(C, dev)
    pad = mode == "pad_fold"
    y = torch.zeros(N, H + 2, W + 2, C, device=dev) if pad else torch.empty(N, H, W, C, device=dev)
    kw = {}
    if res is not None:
        kw["res"] = nhwc(res).to(dev)
    if bn2 is not None:
        st2 = ops.BnState(C, dev)
        kw.update(stats2=stats_of(res).to(dev), bn2=bn2, st2=st2)
    ops.bn_apply(xd, stats, bn, st, y, H, W, N * H * W, relu=True, pad_out=pad, **kw)
    torch.cuda.synchronize()
    if pad:
        assert (nchw(y) - F.pad(y_ref.detach(), (1, 1, 1, 1), mode="replicate")).abs().max() < 1e-5
        y_int = y[:, 1:-1, 1:-1]
    else:
        assert (nchw(y) - y_ref.detach()).abs().max() < 1e-5
        y_int = y
    assert torch.allclose(bn["running_mean"].cpu(), rm, atol=1e-6) and torch.allclose(bn["running_var"].cpu(), rv, atol=1e-5)
    assert int(bn["num_batches_tracked"]) == 1

    dx = torch.empty(N, H, W, C, device=dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    gsrc = nhwc(gpad).to(dev) if pad else nhwc(gy).to(dev)
    gout = torch.empty(N, H, W, C, device=dev) if res is not None else None
    ops.bn_backward(gsrc, y_int, xd, bn["weight"], st, dx, dgam, dbet, H, W, relu=True, fold=pad, g_out=gout)
    torch.cuda.synchronize()
    sc = xr.grad.abs().max()
    assert (nchw(dx) - xr.grad).abs().max() < 2e-5 * sc + 1e-6
    assert torch.allclose(dgam.cpu(), gam.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dbet.cpu(), bet.grad, rtol=1e-4, atol=1e-4)
    if mode == "res":
        assert (nchw(gout) - res_r.grad).abs().max() < 1e-6
    if mode == "res_bn2":
        dx2 = torch.empty(N, H, W, C, device=dev)
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.bn_backward(gout, None, kw["res"], bn2["weight"], st2, dx2, dg2, db2, H, W, relu=False)
        torch.cuda.synchronize()
        assert (nchw(dx2) - res_r.grad).abs().max() < 2e-5 * res_r.grad.abs().max() + 1e-6
        assert torch.allclose(dg2.cpu(), g2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,C,H,W,dtype,res", [(4, 64, 160, 160, torch.bfloat16, True), (16, 1024, 20, 64, torch.float32, False),
                                               (2, 2048, 10, 32, torch.bfloat16, True), (3, 512, 33, 47, torch.float32, True)])
def test_bn_dense_fast_path_large_and_wide(dev, N, C, H, W, dtype, res):
    """bn_apply / bn_bwd_apply on dense tensors large enough that a thread walks several rows (the fast path's ring of raw
    operands two rows ahead) and wide enough for channel slabs (C > 32 16-byte groups: blockIdx.y), and bn_bwd_reduce on
    the capped wide-row grid — against batch_norm + relu (+ residual) and its autograd (resnet.py:33-50, 70-89)."""
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(11 + C)
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).to(dtype).float()
    r = torch.randn(N, C, H, W, generator=g).to(dtype).float() if res else None
    bn = bn_dict(C, g, dev)
    xr = x.clone().requires_grad_(True)
    gam = bn["weight"].cpu().clone().requires_grad_(True)
    bet = bn["bias"].cpu().clone().requires_grad_(True)
    out = F.batch_norm(xr, torch.zeros(C), torch.ones(C), gam, bet, training=True, momentum=0.1, eps=1e-5)
    if res:
        out = out + r
    y_ref = F.relu(out)
    gy = torch.randn(N, C, H, W, generator=g).to(dtype).float()
    y_ref.backward(gy)
    xd = nhwc(x).to(dev).to(dtype)
    st = ops.BnState(C, dev)
    y = torch.empty(N, H, W, C, device=dev, dtype=dtype)
    kw = {"res": nhwc(r).to(dev).to(dtype)} if res else {}
    ops.bn_apply(xd, stats_of(x).to(dev), bn, st, y, H, W, N * H * W, relu=True, **kw)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (nchw(y) - y_ref.detach()).abs().max() <= tol * max(1.0, float(y_ref.detach().abs().max()))
    dx = torch.empty(N, H, W, C, device=dev, dtype=dtype)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    # (the mask comes from the reference's activation so that rounding of y at zero cannot flip a gradient)
    yd = nhwc(y_ref.detach()).to(dev).to(dtype)
    ops.bn_backward(nhwc(gy).to(dev).to(dtype), yd, xd, bn["weight"], st, dx, dgam, dbet, H, W, relu=True)
    torch.cuda.synchronize()
    sc = float(xr.grad.abs().max())
    assert float((nchw(dx) - xr.grad).abs().max()) <= (2e-5 if dtype == torch.float32 else 2e-2) * sc + 1e-6
    rt = 1e-4 if dtype == torch.float32 else 2e-2
    assert torch.allclose(dgam.cpu(), gam.grad, rtol=rt, atol=rt * float(gam.grad.abs().max()))
    assert torch.allclose(dbet.cpu(), bet.grad, rtol=rt, atol=rt * float(bet.grad.abs().max()))


def test_bn_bf16_unit(dev):
    """bf16 storage of the BN input/output: errors stay at bf16 rounding level (no mask flips by construction:
    the oracle sees the same bf16-rounded conv output)."""
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(15)
    N, C, H, W = 4, 64, 12, 20
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).bfloat16().float()
    bn = bn_dict(C, g, dev)
    xr = x.clone().requires_grad_(True)
    gam = bn["weight"].cpu().clone().requires_grad_(True)
    bet = bn["bias"].cpu().clone().requires_grad_(True)
    y_ref = F.relu(F.batch_norm(xr, torch.zeros(C), torch.ones(C), gam, bet, training=True, eps=1e-5))
    gy = torch.randn(N, C, H, W, generator=g).bfloat16().float()
    y_ref.backward(gy)
    xd = nhwc(x, torch.bfloat16).to(dev)
    st = ops.BnState(C, dev)
    y = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=dev)
    ops.bn_apply(xd, stats_of(x).to(dev), bn, st, y, H, W, N * H * W, relu=True)
    # use the oracle's own mask (y_ref > 0) positions where the bf16 output did not round to the other side
    dx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_backward(nhwc(gy, torch.bfloat16).to(dev), y, xd, bn["weight"], st, dx, dg, db, H, W, relu=True)
    torch.cuda.synchronize()
    assert (nchw(y) - y_ref.detach()).abs().max() < 2e-2
    assert float((nchw(dx) - xr.grad).norm() / xr.grad.norm()) < 2e-2
    assert float((dg.cpu() - gam.grad).norm() / gam.grad.norm()) < 1e-2
    assert float((db.cpu() - bet.grad).norm() / bet.grad.norm()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool(dev, dtype):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(2)
    x = F.relu(torch.randn(2, 16, 12, 20, generator=g)).to(dtype).float()  # zeros and bf16 ties present
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(y_ref.shape, generator=g).to(dtype).float()
    add = torch.randn(x.shape, generator=g).to(dtype).float()
    y_ref.backward(gy)
    y, idx = ops.maxpool_fwd(nhwc(x, dtype).to(dev))
    dx = ops.maxpool_bwd(nhwc(gy, dtype).to(dev), idx, 12, 20, addend=nhwc(add, dtype).to(dev))
    torch.cuda.synchronize()
    assert (nchw(y) - y_ref.detach()).abs().max() == 0
    tol = 0 if dtype == torch.float32 else 4e-2
    assert (nchw(dx) - (xr.grad + add)).abs().max() <= tol + 1e-6


def test_upcat_pad(dev):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, 8, 5, 7, generator=g)
    b = torch.randn(2, 12, 10, 14, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.pad(torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), br], 1), (1, 1, 1, 1), mode="replicate")
    gp = torch.randn(ref.shape, generator=g)
    ref.backward(gp)
    out = ops.upcat_pad_fwd(nhwc(a).to(dev), nhwc(b).to(dev))
    da, db = ops.upcat_pad_bwd(nhwc(gp).to(dev), 5, 7, 8, 12)
    torch.cuda.synchronize()
    assert (nchw(out) - ref.detach()).abs().max() == 0
    assert (nchw(da) - ar.grad).abs().max() < 1e-5 and (nchw(db) - br.grad).abs().max() < 1e-5
    out2 = ops.upcat_pad_fwd(nhwc(a).to(dev), None)
    torch.cuda.synchronize()
    assert (nchw(out2) - F.pad(F.interpolate(a, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="replicate")).abs().max() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 5, 7, 8, 12), (3, 6, 20, 32, 64), (2, 12, 40, 256, 256), (2, 9, 11, 16, 0)])
def test_upcat_pad_bwd_with_batchnorm_sums(dev, shape, dtype):
    """fs_upcat_pad_bwd_bn == fs_upcat_pad_bwd followed by the first BatchNorm-backward pass (fs_bn_bwd_reduce with the ReLU
    mask): the same masked gradient (bit for bit) and the same (sum g, sum g * xhat)"""
    from fsnet_amd.hip import ops
    N, h, w, Ca, Cb = shape
    g = torch.Generator().manual_seed(11 + Ca)
    gp = torch.randn(N, 2 * h + 2, 2 * w + 2, Ca + Cb, generator=g).to(dev).to(dtype)
    y = torch.randn(N, h, w, Ca, generator=g).clamp_min(0).to(dev).to(dtype)
    c = torch.randn(N, h, w, Ca, generator=g).to(dev).to(dtype)
    st = ops.BnState(Ca, dev, 1)
    st.mean.copy_(torch.randn(Ca, generator=g) * 0.2)
    st.invstd.copy_(torch.rand(Ca, generator=g) + 0.5)
    st.count = float(N * h * w)
    da0, db0 = ops.upcat_pad_bwd(gp, h, w, Ca, Cb)
    sums = torch.zeros(ops.STAT_SLOTS, 2, Ca, dtype=torch.float64, device=dev)
    da1, db1 = ops.upcat_pad_bwd(gp, h, w, Ca, Cb, bn=(y, c, st, sums))
    torch.cuda.synchronize()
    masked = da0.float() * (y.float() > 0)
    assert torch.equal(da1.float(), masked)
    assert (db0 is None and db1 is None) or torch.equal(db0, db1)
    xhat = (c.float().double() - st.mean.double()) * st.invstd.double()
    want = torch.stack([masked.double().sum((0, 1, 2)), (masked.double() * xhat).sum((0, 1, 2))])
    got = sums.sum(0)
    assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_channel_sum(dev):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(4)
    for C, creal in ((16, 16), (16, 12), (64, 64), (256, 256)):
        x = torch.randn(3, 9, 11, C, generator=g)
        out = torch.ones(C, device=dev)
        ops.channel_sum(x.to(dev), out, creal)
        torch.cuda.synchronize()
        ref = x.sum(dim=(0, 1, 2)) + 1
        ref[creal:] = 1
        assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-4)


def test_depth_head_golden_and_grad(dev):
    from fsnet_amd.hip import ops
    gd = np.load(os.path.join(GOLD, "ops.npz"))
    logits = torch.from_numpy(gd["head_logits"])  # [B,16,h,w]; contains |x| > 10 (clamp path)
    bins = torch.from_numpy(gd["head_bins"])
    ld = nhwc(logits).to(dev)
    depth, disp = ops.depth_head_fwd(ld, bins.to(dev), 16, 0.5, 100.0)
    torch.cuda.synchronize()
    assert (depth.cpu() - torch.from_numpy(gd["head_depth"])).abs().max() < 2e-4
    assert (disp.cpu() - torch.from_numpy(gd["head_disp"])).abs().max() < 1e-5
    lr = logits.clone().requires_grad_(True)
    d = O.gather_activation(lr, bins)
    dp = O.depth_to_disp(d, 0.5, 100.0)
    g = torch.Generator().manual_seed(1)
    gd_, gp_ = torch.randn(d.shape, generator=g), torch.randn(d.shape, generator=g)
    (d * gd_ + dp * gp_).sum().backward()
    dl = ops.depth_head_bwd(ld, bins.to(dev), gd_.to(dev), gp_.to(dev), 16, 0.5, 100.0, torch.float32)
    torch.cuda.synchronize()
    assert (nchw(dl) - lr.grad).abs().max() < 1e-4 * lr.grad.abs().max()


def test_depth_head_multi_scale_launch_equals_per_scale(dev):
    """fs_depth_head_fwd_multi / bwd_multi (all decoder scales in one launch) give bit-identical results to the
    per-scale entry points, with missing gradients (None) and ragged sizes"""
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(4)
    bins = torch.logspace(-0.3, 2.0, 16).to(dev)
    shapes = [(2, 24, 40), (2, 12, 20), (2, 6, 10), (2, 3, 5)]
    logits = [(torch.randn(n, h, w, 16, generator=g) * 6).to(dev) for n, h, w in shapes]
    multi = ops.depth_head_fwd_multi(logits, bins, 16, 0.5, 100.0)
    for lg, (depth, disp) in zip(logits, multi):
        d1, p1 = ops.depth_head_fwd(lg, bins, 16, 0.5, 100.0)
        assert torch.equal(depth, d1) and torch.equal(disp, p1)
    gds = [torch.randn(n, 1, h, w, generator=g).to(dev) for n, h, w in shapes]
    gps = [torch.randn(n, 1, h, w, generator=g).to(dev) for n, h, w in shapes]
    gds[1], gps[2] = None, None
    for dt in (torch.float32, torch.bfloat16):
        dls = ops.depth_head_bwd_multi(logits, bins, gds, gps, 16, 0.5, 100.0, dt)
        for lg, a, b, dl in zip(logits, gds, gps, dls):
            assert torch.equal(dl, ops.depth_head_bwd(lg, bins, a, b, 16, 0.5, 100.0, dt))
    torch.cuda.synchronize()


@pytest.mark.parametrize("invert", [False, True])
def test_pose_tail(dev, invert):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(9)
    B, h, w = 3, 6, 20
    x = torch.randn(B, 12, h, w, generator=g) * 3
    xr = x.clone().requires_grad_(True)
    m = xr.mean(3).mean(2)
    o = 0.01 * m.view(-1, 2, 1, 6)
    aa, tr = o[..., :3], o[..., 3:]
    T = O.transformation_from_parameters(aa[:, 0], tr[:, 0], invert=invert)
    gT = torch.randn(B, 4, 4, generator=g)
    (T * gT).sum().backward()
    xd = torch.zeros(B, h, w, 16, device=dev)
    xd[..., :12] = x.permute(0, 2, 3, 1).to(dev)
    aa_d, tr_d, T_d = ops.pose_tail_fwd(xd, 2, invert)
    dx = ops.pose_tail_bwd(xd, gT.to(dev), 2, invert, torch.float32)
    torch.cuda.synchronize()
    assert (aa_d.cpu() - aa.detach()).abs().max() < 1e-7 and (tr_d.cpu() - tr.detach()).abs().max() < 1e-7
    assert (T_d.cpu() - T.detach()).abs().max() < 1e-6
    ref = xr.grad.permute(0, 2, 3, 1)
    assert (dx[..., :12].cpu() - ref).abs().max() < 1e-4 * ref.abs().max() + 1e-9
    assert dx[..., 12:].abs().max() == 0


def _chain_case():
    g = np.load(os.path.join(GOLD, "loss_chain.npz"))
    T_ = lambda a: torch.from_numpy(np.asarray(a))
    data = {("original_image", 0): T_(g["img_0"]), ("original_image", 1): T_(g["img_p"]),
            ("original_image", -1): T_(g["img_m"]), "P2": T_(g["P2"]), "patched_mask": T_(g["patched_mask"])}
    return g, data


def test_photometric_chain_vs_reference_golden(dev):
    """Full fused loss chain (fwd + bwd) against the REAL reference's outputs and gradients."""
    from fsnet_amd.hip import ops
    g, data = _chain_case()
    B, H, W = data["P2"].shape[0], int(g["H"]), int(g["W"])
    depths = [torch.from_numpy(g["depth_%d" % s]) for s in range(4)]
    disps = [O.depth_to_disp(d, 0.5, 100.0) for d in depths]
    Ts, leaves = [], []
    for f, tag in ((1, "p"), (-1, "m")):
        aa = torch.from_numpy(g["aa_" + tag]).requires_grad_(True)
        tr = torch.from_numpy(g["tr_" + tag]).requires_grad_(True)
        leaves.append((aa, tr))
        Ts.append(O.transformation_from_parameters(aa, tr, invert=(f < 0)))
    pl = ops.PhotometricLoss(B, H, W, [0, 1, 2, 3], dev, 0.5, 100.0)
    out = pl.forward(data[("original_image", 0)].to(dev), [data[("original_image", 1)].to(dev), data[("original_image", -1)].to(dev)],
                     data["P2"].to(dev), [t.detach().to(dev).contiguous() for t in Ts], data["patched_mask"].to(dev),
                     [d.to(dev) for d in depths], [d.to(dev) for d in disps], noise_seed=-1)
    d_depth, d_disp, dT = pl.backward()
    torch.cuda.synchronize()
    out = out.cpu()
    assert out.dtype == torch.float64
    assert abs(float(out[8]) - float(g["total_loss"])) < 5e-7
    for s in range(4):
        assert abs(float(out[s]) - float(g["ld_loss_%d" % s])) < 1e-6
        assert abs(float(out[4 + s]) - float(g["ld_smooth_loss_%d" % s])) < 1e-9
    # warped images / overlap masks (outputs[("original_image", f, 0)])
    for f, tag in ((0, "p"), (1, "m")):
        assert (pl.pred[0, f].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["warp0_" + tag])).abs().max() < 2e-5
        assert (pl.ov[0, f].cpu().bool().numpy() == g["ovmask0_" + tag]).mean() > 0.9995
    # gradients: d depth includes the disp (smoothness) path in the reference -> add it here the same way
    for s in range(4):
        d = depths[s].clone().requires_grad_(True)
        O.depth_to_disp(d, 0.5, 100.0).backward(d_disp[s].cpu())
        got = d_depth[s].cpu() + d.grad
        ref = torch.from_numpy(g["gdepth_%d" % s])
        assert float((got - ref).norm() / ref.norm()) < 3e-3, s
    # pose gradients through dT -> (axisangle, translation)
    for i, tag in enumerate(("p", "m")):
        aa, tr = leaves[i]
        (Ts[i] * dT[i].cpu()).sum().backward()
        for got, key in ((aa.grad, "gaa_" + tag), (tr.grad, "gtr_" + tag)):
            ref = torch.from_numpy(g[key])
            assert (got - ref).abs().max() < 5e-3 * ref.abs().max() + 1e-9, key


def test_photometric_chain_vs_oracle_no_mask(dev):
    """patched_mask=None path and a different geometry, against the oracle."""
    from fsnet_amd.hip import ops
    B, H, W = 2, 32, 64
    data = O.synthetic_batch(B, H, W, seed=8)
    del data["patched_mask"]
    g = torch.Generator().manual_seed(4)
    outputs, leaves = {}, {}
    for s in range(4):
        d = (3 + 20 * torch.rand(B, 1, H >> s, W >> s, generator=g)).requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        outputs[("cam_T_cam", f)] = data[("relative_pose", f)]
    total, ld = O.photometric_loss(outputs, data)
    total.backward()
    pl = ops.PhotometricLoss(B, H, W, [0, 1, 2, 3], dev, 0.5, 100.0)
    out = pl.forward(data[("original_image", 0)].to(dev), [data[("original_image", 1)].to(dev), data[("original_image", -1)].to(dev)],
                     data["P2"].to(dev), [data[("relative_pose", 1)].to(dev), data[("relative_pose", -1)].to(dev)], None,
                     [leaves[s].detach().to(dev) for s in range(4)], [outputs[("disp", s)].detach().to(dev) for s in range(4)])
    d_depth, d_disp, dT = pl.backward()
    torch.cuda.synchronize()
    assert abs(float(out[8].cpu()) - float(total)) < 5e-7
    for s in range(4):
        d = leaves[s].detach().clone().requires_grad_(True)
        O.depth_to_disp(d, 0.5, 100.0).backward(d_disp[s].cpu())
        got = d_depth[s].cpu() + d.grad
        assert float((got - leaves[s].grad).norm() / leaves[s].grad.norm()) < 2e-3


def test_adam_and_clip(dev):
    from fsnet_amd.hip import ops
    g = torch.Generator().manual_seed(6)
    n = 100003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 3
    m = torch.zeros(n); v = torch.zeros(n)
    pd, gd, md, vd = p.to(dev), gr.to(dev), m.to(dev), v.to(dev)
    ss = torch.zeros(1, dtype=torch.float64, device=dev)
    for step in (1, 2, 3):
        norm, clipped = O.clip_grad_norm([gr], 35.0)
        O.adam_step(p, clipped[0], m, v, step, lr=1e-3, weight_decay=1e-5)
        ss.zero_()
        ops.sumsq(gd, ss)
        ops.adam_step(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 1e-5, step, max_norm=35.0, sumsq_buf=ss)
        torch.cuda.synchronize()
        assert abs(float(ss.sqrt()) - float(norm)) < 1e-3 * float(norm)
        assert (pd.cpu() - p).abs().max() < 2e-6


def test_motion_mask_and_pose_term_vs_reference_golden(dev):
    """MonoDepth2Decoder.loss with a precomputed motion mask (no identity auto-mask; gradient scaled by 1 - mask) and
    pose_loss_weight > 0 (mean |relative_pose - cam_T_cam|) against the REAL decoder's loss and gradients"""
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.monodepth.networks.utils.monodepth_utils import transformation_from_parameters
    from fsnet_amd.vision_base.utils.builder import build
    from tests.test_oracle_golden import loss_options_case
    g = np.load(os.path.join(GOLD, "loss_options.npz"))
    H, W = int(g["H"]), int(g["W"])
    RT.tie_noise = False
    head = build(**meta_arch_cfg(H, W, with_pose=True)).head.to(dev)
    head.pose_loss_weight = float(g["pose_loss_weight"])
    data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in loss_options_case(g).items()}
    outputs, leaves, pose = {}, {}, {}
    for s in range(4):
        d = torch.from_numpy(g["depth_%d" % s]).to(dev).requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f, tag in ((1, "p"), (-1, "m")):
        aa = torch.from_numpy(g["aa_" + tag]).to(dev).requires_grad_(True)
        tr = torch.from_numpy(g["tr_" + tag]).to(dev).requires_grad_(True)
        pose[tag] = (aa, tr)
        outputs[("cam_T_cam", f)] = transformation_from_parameters(aa, tr, invert=(f < 0))
    res = head.loss(outputs, data)
    res["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(res["loss"].detach()) - float(g["total_loss"])) < 2e-6 * float(g["total_loss"])
    assert abs(float(res["loss_dict"]["pose_loss"]) - float(g["ld_pose_loss"])) < 1e-6
    for s in range(4):
        assert abs(float(res["loss_dict"]["loss/%d" % s]) - float(g["ld_loss_%d" % s])) < 2e-6 * float(g["ld_loss_%d" % s])
        ref = torch.from_numpy(g["gdepth_%d" % s])
        assert float((leaves[s].grad.cpu() - ref).norm() / ref.norm()) < 3e-3, s
        assert int((head._pl.sel[s] < 2).sum()) == 0          # 2, 3: a reprojection term; 4: the constant 100
    for tag in ("p", "m"):
        for got, key in ((pose[tag][0].grad, "gaa_" + tag), (pose[tag][1].grad, "gtr_" + tag)):
            ref = torch.from_numpy(g[key])
            assert (got.cpu() - ref).abs().max() < 1e-2 * ref.abs().max() + 1e-9, key


def test_overlapped_mask_off_vs_reference_golden(dev):
    """MonoDepth2Decoder.loss with overlapped_mask=False (configs/multi_dataset_example, nusc_wpose_example) on the
    fused kernels against the REAL decoder: loss per scale, depth and pose gradients — 14.6 % of the samples of this
    case leave the source frames, which is where the option acts"""
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.monodepth.networks.utils.monodepth_utils import transformation_from_parameters
    from fsnet_amd.vision_base.utils.builder import build
    from tests.test_oracle_golden import no_overlap_case
    g = np.load(os.path.join(GOLD, "no_overlap_mask.npz"))
    H, W = int(g["H"]), int(g["W"])
    RT.tie_noise = False
    try:
        cfg = meta_arch_cfg(H, W, with_pose=True)
        cfg.head_cfg.overlapped_mask = False
        head = build(**cfg).head.to(dev)
        data, depths, poses = no_overlap_case(H, W)
        data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
        outputs, leaves, pl = {}, {}, {}
        for s in range(4):
            d = depths[s].to(dev).requires_grad_(True)
            leaves[s] = d
            outputs[("depth", s, s)] = d
            outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
        for f, tag in ((1, "p"), (-1, "m")):
            aa, tr = poses[f][0].to(dev).requires_grad_(True), poses[f][1].to(dev).requires_grad_(True)
            pl[tag] = (aa, tr)
            outputs[("cam_T_cam", f)] = transformation_from_parameters(aa, tr, invert=(f < 0))
        res = head.loss(outputs, data)
        res["loss"].backward()
        torch.cuda.synchronize()
        assert not any(k[0] == "overlapped_mask" for k in outputs if isinstance(k, tuple))
        assert abs(float(res["loss"].detach()) - float(g["total_loss"])) < 5e-6 * float(g["total_loss"])
        assert abs(float(res["loss"].detach()) - float(g["total_loss_masked"])) > 1e-3      # not the masked value
        for s in range(4):
            assert abs(float(res["loss_dict"]["loss/%d" % s]) - float(g["ld_loss_%d" % s])) < 5e-6 * float(g["ld_loss_%d" % s])
            ref = torch.from_numpy(g["gdepth_%d" % s])
            assert float((leaves[s].grad.cpu() - ref).norm() / ref.norm()) < 1e-2, s
            assert int((head._pl.sel[s] == 4).sum()) == 0                  # no constant-100 selections without the mask
        for tag in ("p", "m"):
            for got, key in ((pl[tag][0].grad, "gaa_" + tag), (pl[tag][1].grad, "gtr_" + tag)):
                ref = torch.from_numpy(g[key])
                assert (got.cpu() - ref).abs().max() < 1e-2 * ref.abs().max() + 1e-9, key
    finally:
        RT.tie_noise = True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,W,G,keep_y,with_add", [(2, 8, 12, 1, True, True), (4, 12, 20, 2, False, False),
                                                     (3, 96, 64, 1, True, False), (2, 6, 10, 2, True, True)])
def test_stem_batchnorm_relu_maxpool_in_one_pass(dev, dtype, N, H, W, G, keep_y, with_add):
    """fs_bn_apply with FsBnApplyArgs.pool_y (BatchNorm + ReLU + MaxPool2d(3, 2, 1) of the stem, resnet.py:201-206) ==
    fs_bn_apply followed by fs_maxpool_fwd bit for bit (activation, pooled tensor, argmax codes, saved and running
    statistics); its backward with the pooling gradient gathered inside both BatchNorm-backward passes
    (FsBnBwdArgs.pool_dy; the ReLU mask from the stored activation or, where it was never stored, from the raw
    convolution output) == fs_maxpool_bwd + the two passes on the materialised gradient, and == torch autograd."""
    import copy
    import torch.nn.functional as F
    from fsnet_amd.engine.nets import bn_tensors
    from fsnet_amd.hip import ops
    Cc = 64
    g = torch.Generator().manual_seed(11 + H)
    x = torch.randn(N, H, W, Cc, generator=g).to(dev).to(dtype)
    bn0 = torch.nn.BatchNorm2d(Cc).to(dev)
    bn0.weight.data.copy_(torch.rand(Cc, generator=g) + 0.5); bn0.bias.data.copy_(torch.rand(Cc, generator=g) - 0.5)
    n = N // G
    stats = torch.zeros(G, 8, 2, Cc, dtype=torch.float64, device=dev)
    for gi in range(G):
        v = x[gi * n:(gi + 1) * n].double()
        stats[gi, 0, 0] = v.sum((0, 1, 2)); stats[gi, 0, 1] = (v * v).sum((0, 1, 2))
    stats_arg = stats if G > 1 else stats[0]
    count = n * H * W
    # separate passes
    m_a, st_a = copy.deepcopy(bn0), ops.BnState(Cc, dev, G)
    y_a = torch.empty_like(x)
    ops.bn_apply(x, stats_arg, bn_tensors(m_a), st_a, y_a, H, W, count, relu=True, groups=G)
    p_a, idx_a = ops.maxpool_fwd(y_a)
    # one pass
    m_b, st_b = copy.deepcopy(bn0), ops.BnState(Cc, dev, G)
    y_b = torch.full_like(x, 7.0) if keep_y else None
    p_b = torch.empty_like(p_a); idx_b = torch.empty_like(idx_a)
    ops.bn_apply(x, stats_arg, bn_tensors(m_b), st_b, y_b, H, W, count, relu=True, groups=G, pool=(p_b, idx_b))
    torch.cuda.synchronize()
    assert torch.equal(p_a, p_b) and torch.equal(idx_a, idx_b)
    if keep_y:
        assert torch.equal(y_a, y_b)
    assert torch.equal(st_a.mean, st_b.mean) and torch.equal(st_a.invstd, st_b.invstd)
    assert torch.equal(m_a.running_mean, m_b.running_mean) and torch.equal(m_a.running_var, m_b.running_var)
    assert int(m_a.num_batches_tracked) == int(m_b.num_batches_tracked) == G
    # backward
    dpool = torch.randn(p_a.shape, generator=g).to(dev).to(dtype)
    add = torch.randn(x.shape, generator=g).to(dev).to(dtype) if with_add else None
    gamma = bn0.weight.data
    d0 = ops.maxpool_bwd(dpool, idx_a, H, W, addend=add)
    dx_a = torch.empty_like(x)
    dg_a, db_a = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ops.bn_backward(d0, y_a, x, gamma, st_a, dx_a, dg_a, db_a, H, W, relu=True)
    dx_b = torch.empty_like(x)
    dg_b, db_b = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ops.bn_backward(add, y_b, x, gamma, st_b, dx_b, dg_b, db_b, H, W, relu=True, pool=(dpool, idx_b, bn0.bias.data))
    torch.cuda.synchronize()
    # (the separate path rounds the gathered gradient to the storage type before the BatchNorm passes read it)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    scale = float(dx_a.float().abs().max())
    assert float((dx_a.float() - dx_b.float()).abs().max()) <= tol * scale
    assert float((dg_a - dg_b).abs().max()) <= tol * float(dg_a.abs().max()) + 1e-6
    assert float((db_a - db_b).abs().max()) <= tol * float(db_a.abs().max()) + 1e-6
    if dtype == torch.float32:
        # torch autograd of the same chain, group by group (G calls of the module)
        for gi in range(G):
            xr = x[gi * n:(gi + 1) * n].permute(0, 3, 1, 2).clone().requires_grad_(True)
            ref = copy.deepcopy(bn0).train()
            yy = F.relu(ref(xr))
            pp = F.max_pool2d(yy, 3, 2, 1)
            loss = (pp * dpool[gi * n:(gi + 1) * n].permute(0, 3, 1, 2)).sum()
            if with_add:
                loss = loss + (yy * add[gi * n:(gi + 1) * n].permute(0, 3, 1, 2)).sum()
            loss.backward()
            got = dx_b[gi * n:(gi + 1) * n].permute(0, 3, 1, 2)
            assert float((got - xr.grad).abs().max()) <= 2e-4 * float(xr.grad.abs().max())
