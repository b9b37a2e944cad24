"""Value-level parity of the BatchNorm-fold paths and of every tile configuration of the 32x32-tile kernel against
torch's fp32 conv2d / autograd on the CPU (the ATen ops the reference calls:
vision_base/networks/models/backbone/resnet.py:33-50, conv1 -> bn1 -> relu -> conv2 and its backward).

  * forward with the operand prologue (FsConvArgs.pro_mode = 1): reference = conv2d(round_bf16(relu(scale*x + shift)), w)
    with the zero padding applied AFTER the transform;
  * weight gradient with the same prologue on its input operand (FsWgradArgs.pro_a);
  * data gradient whose epilogue derives the ReLU mask from scale*c + shift (bnb_scale) and accumulates the
    BatchNorm-backward sums;
at sizes where the benchmark step runs them (the launches the 32x32-tile kernel takes by its own choice), and the
32x32-tile kernel's three tile configurations x {fp32, bf16} x the epilogue combinations the networks use, forced through
FsConvArgs.force_impl (hip.conv.FORCE_3X3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


def _affine(x_nhwc, scale, shift, G, relu=True):
    """per-group scale * x + shift (+ ReLU), rounded once to bf16 — fp32 multiply, then add, as the kernels do
    (the library is built with -ffp-contract=off)"""
    N = x_nhwc.shape[0]
    C = x_nhwc.shape[-1]
    out = torch.empty_like(x_nhwc)
    n = N // G
    for g in range(G):
        v = x_nhwc[g * n:(g + 1) * n] * scale[g * C:(g + 1) * C] + shift[g * C:(g + 1) * C]
        out[g * n:(g + 1) * n] = torch.relu(v) if relu else v
    return _bf(out)


FOLD_CASES = [
    # Ci, Co, N, H, W, groups — shapes whose launch the 32x32-tile kernel takes by its own choice (ConvOp.can_fold_input)
    (64, 64, 12, 48, 160, 1),      # layer 1 of the depth encoder at the benchmark batch
    (128, 128, 24, 24, 80, 2),     # layer 2 of the stacked pose pass: two statistics groups
]


@pytest.mark.parametrize("case", FOLD_CASES)
def test_forward_with_batchnorm_prologue_matches_conv_of_normalised_input(dev, case):
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G = case
    g = torch.Generator().manual_seed(100 + Ci + N)
    x = _bf(torch.randn(N, H, W, Ci, generator=g))                  # the RAW output of conv1
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    scale = torch.rand(G * Ci, generator=g) + 0.5
    shift = torch.randn(G * Ci, generator=g) * 0.3
    xn = _affine(x, scale, shift, G)
    ref = F.conv2d(xn.permute(0, 3, 1, 2), w, None, padding=1)     # zero padding of the normalised tensor
    op = ConvOp(Ci, Co, 3, 3, 1, 1, torch.bfloat16, dev)
    assert op.can_fold_input(N, H, W), "the case must be one the benchmark folds"
    op.pack(w.to(dev).contiguous())
    st = ops.BnState(Ci, dev, G, affine=True)
    st.scale.copy_(scale); st.shift.copy_(shift)
    xd = x.to(dev).bfloat16()
    scale_y = ref.abs().max().item()
    n = N // G
    for out_f32, tol in ((False, 6e-3), (True, 2e-3)):            # templated EP_STATS epilogue / run-time flags
        stats = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
        y = op.forward(xd, stats=stats, stat_groups=G, pro=(st, True), out_f32=out_f32)
        torch.cuda.synchronize()
        got = y.float().permute(0, 3, 1, 2).cpu()
        assert (got - ref).abs().max().item() <= tol * scale_y, (out_f32, (got - ref).abs().max().item() / scale_y)
        for gi in range(G):
            r = ref[gi * n:(gi + 1) * n].double()
            s = stats[gi].sum(0).cpu()
            assert torch.allclose(s[0], r.sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-3 * (r ** 2).sum(dim=(0, 2, 3)).max().sqrt().item())
            assert torch.allclose(s[1], (r ** 2).sum(dim=(0, 2, 3)), rtol=3e-3)
    # without the ReLU flag the prologue is the affine map alone
    xn2 = _affine(x, scale, shift, G, relu=False)
    ref2 = F.conv2d(xn2.permute(0, 3, 1, 2), w, None, padding=1)
    stats = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
    y2 = op.forward(xd, stats=stats, stat_groups=G, pro=(st, False), out_f32=True)
    torch.cuda.synchronize()
    assert (y2.permute(0, 3, 1, 2).cpu() - ref2).abs().max().item() <= 2e-3 * ref2.abs().max().item()


@pytest.mark.parametrize("case", FOLD_CASES)
def test_weight_gradient_with_batchnorm_prologue_matches_autograd(dev, case):
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G = case
    g = torch.Generator().manual_seed(200 + Ci + N)
    x = _bf(torch.randn(N, H, W, Ci, generator=g))
    scale = torch.rand(G * Ci, generator=g) + 0.5
    shift = torch.randn(G * Ci, generator=g) * 0.3
    xn = _affine(x, scale, shift, G)
    gy = _bf(torch.randn(N, H, W, Co, generator=g))
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(xn.permute(0, 3, 1, 2), w, None, padding=1).backward(gy.permute(0, 3, 1, 2))
    op = ConvOp(Ci, Co, 3, 3, 1, 1, torch.bfloat16, dev)
    st = ops.BnState(Ci, dev, G, affine=True)
    st.scale.copy_(scale); st.shift.copy_(shift)
    dw = torch.zeros(Co, Ci, 3, 3, device=dev)
    op.wgrad(gy.to(dev).bfloat16(), x.to(dev).bfloat16(), dw, pro=(st, True))
    torch.cuda.synchronize()
    wscale = w.grad.abs().max().item()
    assert (dw.cpu() - w.grad).abs().max().item() <= 2e-3 * wscale
    # and it accumulates
    op.wgrad(gy.to(dev).bfloat16(), x.to(dev).bfloat16(), dw, pro=(st, True))
    torch.cuda.synchronize()
    assert (dw.cpu() - 2 * w.grad).abs().max().item() <= 4e-3 * wscale


@pytest.mark.parametrize("case", FOLD_CASES)
def test_data_gradient_with_derived_relu_mask_matches_autograd(dev, case):
    """dgrad(bn_fuse=(c, st, sums), mask_bn=True): dx = conv_transpose(dy) where scale*c + shift > 0, else 0, and
    sums = (sum dx, sum dx * xhat), xhat = (c - mean) * invstd, per statistics group"""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G = case
    g = torch.Generator().manual_seed(300 + Ci + N)
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    dy = _bf(torch.randn(N, H, W, Co, generator=g))
    c = _bf(torch.randn(N, H, W, Ci, generator=g))                  # raw output of the convolution in front of the BatchNorm
    scale = torch.rand(G * Ci, generator=g) + 0.5
    shift = torch.randn(G * Ci, generator=g) * 0.3
    mean = torch.randn(G * Ci, generator=g) * 0.1
    invstd = torch.rand(G * Ci, generator=g) + 0.5
    dx_ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), w, None, padding=1).permute(0, 2, 3, 1)   # [N,H,W,Ci] fp32
    n = N // G
    keep = torch.empty(N, H, W, Ci, dtype=torch.bool)
    xhat = torch.empty(N, H, W, Ci)
    for gi in range(G):
        sl = slice(gi * n, (gi + 1) * n)
        keep[sl] = (c[sl] * scale[gi * Ci:(gi + 1) * Ci] + shift[gi * Ci:(gi + 1) * Ci]) > 0
        xhat[sl] = (c[sl] - mean[gi * Ci:(gi + 1) * Ci]) * invstd[gi * Ci:(gi + 1) * Ci]
    g_ref = torch.where(keep, dx_ref, torch.zeros(()))
    op = ConvOp(Ci, Co, 3, 3, 1, 1, torch.bfloat16, dev)
    op.pack(w.to(dev).contiguous())
    st = ops.BnState(Ci, dev, G, affine=True)
    st.scale.copy_(scale); st.shift.copy_(shift); st.mean.copy_(mean); st.invstd.copy_(invstd)
    st.count = float(n * H * W)
    sums = torch.zeros(G, 8, 2, Ci, dtype=torch.float64, device=dev)
    got = op.dgrad(dy.to(dev).bfloat16(), H, W, bn_fuse=(c.to(dev).bfloat16(), st, sums), mask_bn=True)
    torch.cuda.synchronize()
    gscale = dx_ref.abs().max().item()
    assert (got.float().cpu() - g_ref).abs().max().item() <= 1e-2 * gscale
    assert ((got.float().cpu() != 0) <= keep).all()                # nothing leaks through the mask
    for gi in range(G):
        sl = slice(gi * n, (gi + 1) * n)
        s = sums[gi].sum(0).cpu()
        r0 = g_ref[sl].double().sum(dim=(0, 1, 2))
        r1 = (g_ref[sl].double() * xhat[sl].double()).sum(dim=(0, 1, 2))
        big = max(r0.abs().max().item(), r1.abs().max().item())
        # (the kernel's sums come from the fp32 accumulators before the bf16 store)
        assert (s[0] - r0).abs().max().item() <= 2e-3 * big + 1e-3 * (g_ref[sl].double() ** 2).sum().sqrt().item() / Ci ** 0.5
        assert (s[1] - r1).abs().max().item() <= 2e-3 * big + 1e-3 * (g_ref[sl].double() ** 2).sum().sqrt().item() / Ci ** 0.5


# ---------------------------------------------------------------------------------------------------------------
# every tile configuration of the 32x32-tile kernel, both element types, the epilogues the networks use
# ---------------------------------------------------------------------------------------------------------------
T32_SHAPES = [
    # Ci, Co, N, H, W
    (64, 64, 2, 16, 32),
    (128, 64, 3, 12, 20),      # ragged pixel tiles
    (64, 128, 2, 9, 33),       # odd sizes: partial tiles in both directions
]


def _nhwc(x, dev, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("cfg", [1, 2, 3])
@pytest.mark.parametrize("shape", T32_SHAPES)
def test_t32_tile_configurations_against_conv2d(dev, monkeypatch, shape, cfg, dtype):
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    from fsnet_amd.hip import conv as _conv
    monkeypatch.setattr(_conv, "FORCE_3X3", 1 + cfg)        # FsConvArgs.force_impl 2-4: the 32x32-tile kernel, configuration cfg
    Ci, Co, N, H, W = shape
    lo = dtype == torch.bfloat16
    g = torch.Generator().manual_seed(17 * cfg + Ci + Co + H)
    rnd = (lambda t: _bf(t)) if lo else (lambda t: t)
    x = rnd(torch.randn(N, Ci, H, W, generator=g))
    w = rnd(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    b = torch.randn(Co, generator=g)
    gy = rnd(torch.randn(N, Co, H, W, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, padding=1)
    y_ref.backward(gy)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dtype, dev)
    op.pack(w.to(dev).contiguous())
    plan = op.plan_3x3(N, H, W, forward=True)
    assert plan["kernel"] == "t32", plan                            # the forced configuration is what runs
    assert (plan["pix"], plan["co"]) == {1: (128, 64), 2: (128, 32), 3: (256, 32)}[cfg], plan
    xd, gyd = _nhwc(x, dev, dtype), _nhwc(gy, dev, dtype)
    tol = 2e-5 if not lo else 2e-3
    ys = y_ref.detach().abs().max().item()
    bias = b.to(dev)

    # forward: statistics only (encoder), bias + statistics (decoder), bias + ReLU (pose decoder), fp32 output
    stats = torch.zeros(8, 2, Co, dtype=torch.float64, device=dev)
    y = op.forward(xd, stats=stats, out_f32=True)
    torch.cuda.synchronize()
    assert (y.permute(0, 3, 1, 2).cpu() - y_ref.detach()).abs().max().item() <= tol * ys
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0], y_ref.detach().double().sum(dim=(0, 2, 3)), rtol=1e-3, atol=1e-3 * ys * (N * H * W) ** 0.5)
    assert torch.allclose(s[1], (y_ref.detach().double() ** 2).sum(dim=(0, 2, 3)), rtol=2e-3)
    stats.zero_()
    y = op.forward(xd, stats=stats)                                 # templated EP_STATS, output in the compute dtype
    torch.cuda.synchronize()
    assert (y.float().permute(0, 3, 1, 2).cpu() - y_ref.detach()).abs().max().item() <= (tol if not lo else 6e-3) * ys
    stats.zero_()
    yb = op.forward(xd, bias=bias, stats=stats, out_f32=True)
    yr = op.forward(xd, bias=bias, relu=True, out_f32=True)
    torch.cuda.synchronize()
    ref_b = y_ref.detach() + b.view(1, -1, 1, 1)
    assert (yb.permute(0, 3, 1, 2).cpu() - ref_b).abs().max().item() <= tol * ref_b.abs().max().item()
    assert (yr.permute(0, 3, 1, 2).cpu() - ref_b.clamp_min(0)).abs().max().item() <= tol * ref_b.abs().max().item()
    assert torch.allclose(stats.sum(0)[0].cpu(), ref_b.double().sum(dim=(0, 2, 3)), rtol=1e-3, atol=1e-3 * ys * (N * H * W) ** 0.5)

    # data gradient: plain, + addend, ReLU mask, mask + BatchNorm-backward sums, addend + mask + sums, derived mask + sums
    tg = 2e-5 if not lo else 1e-2
    dx_ref = xr.grad
    gs = dx_ref.abs().max().item()
    plan_d = op.plan_3x3(N, H, W, forward=False)
    assert plan_d["kernel"] == "t32", plan_d
    dx = op.dgrad(gyd, H, W)
    torch.cuda.synchronize()
    assert (dx.float().permute(0, 3, 1, 2).cpu() - dx_ref).abs().max().item() <= tg * gs
    add = rnd(torch.randn(N, Ci, H, W, generator=g))
    yact = rnd(torch.randn(N, Ci, H, W, generator=g))
    cprev = rnd(torch.randn(N, Ci, H, W, generator=g))
    mean = torch.randn(Ci, generator=g) * 0.1
    invstd = torch.rand(Ci, generator=g) + 0.5
    scale = torch.rand(Ci, generator=g) + 0.5
    shift = torch.randn(Ci, generator=g) * 0.3
    addd, yd, cd = _nhwc(add, dev, dtype), _nhwc(yact, dev, dtype), _nhwc(cprev, dev, dtype)
    dxa = op.dgrad(gyd, H, W, addend=addd)
    dxm = op.dgrad(gyd, H, W, mask=yd)
    torch.cuda.synchronize()
    assert (dxa.float().permute(0, 3, 1, 2).cpu() - (dx_ref + add)).abs().max().item() <= tg * (dx_ref + add).abs().max().item()
    ref_m = torch.where(yact > 0, dx_ref, torch.zeros(()))
    assert (dxm.float().permute(0, 3, 1, 2).cpu() - ref_m).abs().max().item() <= tg * gs
    st = ops.BnState(Ci, dev, 1, affine=True)
    st.mean.copy_(mean); st.invstd.copy_(invstd); st.scale.copy_(scale); st.shift.copy_(shift)
    st.count = float(N * H * W)
    xh = (cprev - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)

    def check_sums(sums, gref):
        s = sums.sum(0).cpu()
        r0 = gref.double().sum(dim=(0, 2, 3)); r1 = (gref.double() * xh.double()).sum(dim=(0, 2, 3))
        big = max(r0.abs().max().item(), r1.abs().max().item())
        eps = (1e-5 if not lo else 2e-3) * big + (1e-6 if not lo else 1e-3) * gref.double().norm().item() / Ci ** 0.5
        assert (s[0] - r0).abs().max().item() <= eps and (s[1] - r1).abs().max().item() <= eps

    for kw, gref in (
            (dict(mask=yd), ref_m),
            (dict(mask=yd, addend=addd), torch.where(yact > 0, dx_ref + add, torch.zeros(()))),
            (dict(mask_bn=True), torch.where(cprev * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) > 0, dx_ref, torch.zeros(())))):
        sums = torch.zeros(8, 2, Ci, dtype=torch.float64, device=dev)
        d = op.dgrad(gyd, H, W, bn_fuse=(cd, st, sums), **kw)
        torch.cuda.synchronize()
        assert (d.float().permute(0, 3, 1, 2).cpu() - gref).abs().max().item() <= tg * max(gs, gref.abs().max().item()), kw.keys()
        check_sums(sums, gref)

    # operand prologue on this configuration (forward; bf16 and fp32)
    stp = ops.BnState(Ci, dev, 1, affine=True)
    stp.scale.copy_(scale); stp.shift.copy_(shift)
    xn = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xn = rnd(xn)
    ref_p = F.conv2d(xn, w, None, padding=1)
    stats.zero_()
    yp = op.forward(xd, stats=stats, pro=(stp, True), out_f32=True)
    torch.cuda.synchronize()
    assert (yp.permute(0, 3, 1, 2).cpu() - ref_p).abs().max().item() <= tol * ref_p.abs().max().item()


def test_fold_survives_the_t32_switch(dev, monkeypatch):
    """every launch forced onto the 16x16-tile kernel (FsConvArgs.force_impl = 1; until round 5 the FSNET_AMD_T32=0 switch) with
    the default BatchNorm fold: launches that carry a prologue or a derived mask still run (they returned
    FS_EINVAL and the step raised)"""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    from fsnet_amd.hip import conv as _conv
    monkeypatch.setattr(_conv, "FORCE_3X3", 1)              # FsConvArgs.force_impl 1: the 16x16-tile kernel
    Ci = Co = 64
    N, H, W = 12, 48, 160
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(N, H, W, Ci, generator=g))
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / 24)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    op = ConvOp(Ci, Co, 3, 3, 1, 1, torch.bfloat16, dev)
    op.pack(w.to(dev).contiguous())
    st = ops.BnState(Ci, dev, 1, affine=True)
    st.scale.copy_(scale); st.shift.copy_(shift)
    stats = torch.zeros(8, 2, Co, dtype=torch.float64, device=dev)
    y = op.forward(x.to(dev).bfloat16(), stats=stats, pro=(st, True), out_f32=True)
    torch.cuda.synchronize()
    ref = F.conv2d(_affine(x, scale, shift, 1).permute(0, 3, 1, 2), w, None, padding=1)
    assert (y.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


# ---------------------------------------------------------------------------------------------------------------
# round 4: the prologue derives its coefficients in the kernel (no fs_bn_finalize launch).  (The data gradient's
# counterpart — the second pass of the BatchNorm backward applied to dY while staging it — was removed in round 5.)
# ---------------------------------------------------------------------------------------------------------------
FIN_CASES = [
    # Ci, Co, N, H, W, groups, kernel the launch runs on
    (64, 64, 12, 48, 160, 1, "t32"),
    (128, 128, 24, 24, 80, 2, "t32"),
    (256, 256, 4, 12, 40, 2, "halo"),     # 128 x 16 tiles
    (64, 128, 2, 16, 32, 1, "halo"),      # 128 x 32 tiles
    (512, 512, 12, 6, 20, 1, "halo"),     # layer 4 at the benchmark batch
]


def _spread_slots(per_group, gen):
    """[G][2][C] f64 sums -> [G][8][2][C] with the total spread over the address slots as the epilogues leave it"""
    G, _, C = per_group.shape
    w = torch.rand(G, 8, 1, 1, generator=gen, dtype=torch.float64)
    w = w / w.sum(1, keepdim=True)
    out = per_group.unsqueeze(1) * w
    out[:, 0] += per_group - out.sum(1)            # exact total
    return out


def _bn_dict(C, dev, gen):
    return {"weight": (torch.rand(C, generator=gen) + 0.5).to(dev), "bias": (torch.randn(C, generator=gen) * 0.2).to(dev),
            "running_mean": (torch.randn(C, generator=gen) * 0.1).to(dev), "running_var": (torch.rand(C, generator=gen) + 0.5).to(dev),
            "num_batches_tracked": torch.tensor(3, dtype=torch.int64, device=dev)}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("case", FIN_CASES)
def test_forward_finalises_batchnorm_statistics_in_its_prologue(dev, case, dtype):
    """ConvOp.forward(pro=(st, relu, sums, bn, count, track)): the launch == fs_bn_finalize + the coefficient-array
    prologue, including the saved statistics and the running-statistics update its block 0 makes; and against conv2d of the
    normalised input on the CPU"""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G, kern = case
    lo = dtype == torch.bfloat16
    g = torch.Generator().manual_seed(400 + Ci + N)
    rnd = _bf if lo else (lambda t: t)
    x = rnd(torch.randn(N, H, W, Ci, generator=g) * 1.5 + 0.3)
    w = rnd(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    n = N // G
    per = torch.stack([torch.stack([x[gi * n:(gi + 1) * n].double().sum(dim=(0, 1, 2)),
                                    (x[gi * n:(gi + 1) * n].double() ** 2).sum(dim=(0, 1, 2))]) for gi in range(G)])
    stats_in = _spread_slots(per, g).to(dev)
    count = float(n * H * W)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dtype, dev)
    op.pack(w.to(dev).contiguous())
    assert op.plan_3x3(N, H, W, forward=True, pro_mode=1)["kernel"] == kern
    # reference: the finalize kernel
    gb = torch.Generator().manual_seed(9)
    bn_ref, bn_new = _bn_dict(Ci, dev, gb), _bn_dict(Ci, dev, torch.Generator().manual_seed(9))
    st_ref = ops.BnState(Ci, dev, G, affine=True)
    ops.bn_finalize(stats_in, bn_ref, st_ref, Ci, count, track=True, groups=G)
    st = ops.BnState(Ci, dev, G, affine=True)
    xd = x.to(dev).to(dtype)
    sA = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
    sB = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
    yA = op.forward(xd, stats=sA, stat_groups=G, pro=(st_ref, True), out_f32=True)
    yB = op.forward(xd, stats=sB, stat_groups=G, pro=(st, True, stats_in, bn_new, count, True), out_f32=True)
    torch.cuda.synchronize()
    for a, b in ((st.mean, st_ref.mean), (st.invstd, st_ref.invstd), (st.scale, st_ref.scale), (st.shift, st_ref.shift),
                 (bn_new["running_mean"], bn_ref["running_mean"]), (bn_new["running_var"], bn_ref["running_var"])):
        assert torch.equal(a, b)
    assert int(bn_new["num_batches_tracked"]) == 3 + G and st.count == count
    assert torch.equal(yA, yB)                                      # same coefficients, same kernel arithmetic
    assert torch.equal(sA.sum(1), sB.sum(1)) or float((sA.sum(1) - sB.sum(1)).abs().max()) <= 1e-9 * float(sA.sum(1).abs().max())
    xn = _affine(x, st_ref.scale.cpu(), st_ref.shift.cpu(), G) if lo else torch.cat(
        [torch.relu(x[gi * n:(gi + 1) * n] * st_ref.scale.cpu()[gi * Ci:(gi + 1) * Ci] + st_ref.shift.cpu()[gi * Ci:(gi + 1) * Ci]) for gi in range(G)])
    ref = F.conv2d(xn.permute(0, 3, 1, 2), w, None, padding=1)
    assert (yB.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() <= (2e-3 if lo else 2e-5) * ref.abs().max().item()
    # against the definition: mean / biased variance of x per group
    mean_ref = (per[:, 0] / count).float().flatten()
    var_ref = (per[:, 1] / count - (per[:, 0] / count) ** 2).clamp_min(0)
    assert torch.allclose(st.mean.cpu(), mean_ref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(st.invstd.cpu(), (1.0 / torch.sqrt(var_ref + 1e-5)).float().flatten(), rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [(64, 128, 12, 48, 160, 1), (128, 256, 24, 24, 80, 2), (256, 512, 12, 12, 40, 1),
                                  (64, 128, 3, 17, 31, 1), (64, 64, 2, 9, 9, 1)])
def test_stride2_forward_on_the_halo_kernel(dev, case, dtype):
    """3x3 / stride-2 / pad-1 forward (ResNet stage entries, vision_base/networks/models/backbone/resnet.py:140-160) on the
    LDS-halo kernel's stride-2 variant: output and per-group BatchNorm statistics against conv2d on the CPU — full benchmark
    shapes, odd sizes (ragged tiles, a last row / column whose window hangs over the border), statistics groups"""
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G = case
    lo = dtype == torch.bfloat16
    g = torch.Generator().manual_seed(600 + Ci + H)
    rnd = _bf if lo else (lambda t: t)
    x = rnd(torch.randn(N, Ci, H, W, generator=g))
    w = rnd(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    op = ConvOp(Ci, Co, 3, 3, 2, 1, dtype, dev)
    assert op.halo_f_s2 and op.plan_3x3(N, H, W, forward=True)["kernel"] == "halo"
    op.pack(w.to(dev).contiguous())
    xd = _nhwc(x, dev, dtype)
    stats = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
    y = op.forward(xd, stats=stats, stat_groups=G, out_f32=True)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (N, ref.shape[2], ref.shape[3], Co)
    assert (y.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() <= (2e-3 if lo else 2e-5) * ref.abs().max().item()
    n = N // G
    for gi in range(G):
        r = ref[gi * n:(gi + 1) * n].double()
        s = stats[gi].sum(0).cpu()
        assert torch.allclose(s[0], r.sum(dim=(0, 2, 3)), rtol=1e-3, atol=1e-3 * (r ** 2).sum(dim=(0, 2, 3)).max().sqrt().item())
        assert torch.allclose(s[1], (r ** 2).sum(dim=(0, 2, 3)), rtol=2e-3)
    # the implicit GEMM on the same operands agrees
    import fsnet_amd.hip.conv as CV
    op2 = ConvOp(Ci, Co, 3, 3, 2, 1, dtype, dev)
    op2.halo_f_s2 = False
    op2.pack(w.to(dev).contiguous())
    y2 = op2.forward(xd, out_f32=True)
    torch.cuda.synchronize()
    assert (y2 - y).abs().max().item() <= (1e-3 if lo else 2e-5) * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [(16, 16, 2, 64, 96), (32, 16, 2, 48, 80), (16, 32, 3, 40, 57), (16, 16, 12, 192, 640)])
def test_persistent_one_chunk_kernel_against_conv2d(dev, monkeypatch, case, dtype):
    """conv3x3_p1.hip (the depth decoder's 16 / 32-channel layers, depth_encoder.py:45-63: weights resident, a block walks
    pixel tiles): forward with bias + statistics + fp32 output, bias + ReLU, and the data gradient with addend / ReLU mask /
    BatchNorm-backward sums, against conv2d and autograd on the CPU — forced on for small launches through FsConvArgs.force_impl,
    ragged tiles included, and by its own choice at 192x640"""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W = case
    lo = dtype == torch.bfloat16
    if not lo and Ci > 16:
        pytest.skip("fp32: one 64-byte chunk is 16 channels")
    if H < 192:
        from fsnet_amd.hip import conv as _conv
        monkeypatch.setattr(_conv, "FORCE_3X3", 5)          # FsConvArgs.force_impl 5: this kernel whatever the tile count
    g = torch.Generator().manual_seed(700 + Ci + Co + H)
    rnd = _bf if lo else (lambda t: t)
    x = rnd(torch.randn(N, Ci, H, W, generator=g))
    w = rnd(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5)
    b = torch.randn(Co, generator=g)
    gy = rnd(torch.randn(N, Co, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, b, padding=1)
    y_ref.backward(gy)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dtype, dev)
    op.pack(w.to(dev).contiguous())
    assert op.plan_3x3(N, H, W, forward=True)["kernel"] == "p1"
    # (the data gradient's source is dY: Co channels must fit one 64-byte chunk too, else the other kernels take it)
    assert op.plan_3x3(N, H, W, forward=False)["kernel"] == ("p1" if op.Co_p * (2 if lo else 4) <= 64 else "halo")
    xd, gyd = _nhwc(x, dev, dtype), _nhwc(gy, dev, dtype)
    bias = torch.zeros(op.Co_p, device=dev); bias[:Co] = b.to(dev)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(xd, bias=bias, stats=stats, out_f32=True)
    yr = op.forward(xd, bias=bias, relu=True)
    torch.cuda.synchronize()
    tol = 2e-3 if lo else 2e-5
    ys = y_ref.detach().abs().max().item()
    assert (y.permute(0, 3, 1, 2).cpu() - y_ref.detach()).abs().max().item() <= tol * ys
    assert (yr.float().permute(0, 3, 1, 2).cpu() - y_ref.detach().clamp_min(0)).abs().max().item() <= (6e-3 if lo else 2e-5) * ys
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0, :Co], y_ref.detach().double().sum(dim=(0, 2, 3)), rtol=1e-3, atol=1e-3 * ys * (N * H * W) ** 0.5)
    assert torch.allclose(s[1, :Co], (y_ref.detach().double() ** 2).sum(dim=(0, 2, 3)), rtol=2e-3)
    # data gradient (dY has Co channels: one chunk; output Ci channels)
    tg = 1e-2 if lo else 2e-5
    dx_ref = xr.grad
    dx = op.dgrad(gyd, H, W)
    torch.cuda.synchronize()
    assert (dx.float().permute(0, 3, 1, 2).cpu() - dx_ref).abs().max().item() <= tg * dx_ref.abs().max().item()
    add, yact, cprev = (rnd(torch.randn(N, Ci, H, W, generator=g)) for _ in range(3))
    st = ops.BnState(Ci, dev, 1)
    mean, invstd = torch.randn(Ci, generator=g) * 0.1, torch.rand(Ci, generator=g) + 0.5
    st.mean.copy_(mean); st.invstd.copy_(invstd); st.count = float(N * H * W)
    sums = torch.zeros(8, 2, Ci, dtype=torch.float64, device=dev)
    d = op.dgrad(gyd, H, W, addend=_nhwc(add, dev, dtype), mask=_nhwc(yact, dev, dtype), bn_fuse=(_nhwc(cprev, dev, dtype), st, sums))
    torch.cuda.synchronize()
    gref = torch.where(yact > 0, dx_ref + add, torch.zeros(()))
    assert (d.float().permute(0, 3, 1, 2).cpu() - gref).abs().max().item() <= tg * gref.abs().max().item()
    xh = (cprev - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    ss = sums.sum(0).cpu()
    r0, r1 = gref.double().sum(dim=(0, 2, 3)), (gref.double() * xh.double()).sum(dim=(0, 2, 3))
    big = max(r0.abs().max().item(), r1.abs().max().item())
    eps = (2e-3 if lo else 1e-5) * big + (1e-3 if lo else 1e-6) * gref.double().norm().item() / Ci ** 0.5
    assert (ss[0] - r0).abs().max().item() <= eps and (ss[1] - r1).abs().max().item() <= eps


def test_persistent_one_chunk_kernel_statistic_groups(dev, monkeypatch):
    """conv3x3_p1.hip with the batch split into two BatchNorm invocations (stacked calls, resnet.py train-mode BatchNorm per
    call): forward statistics and the data gradient's BatchNorm-backward sums land in the group of the tile's image"""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    from fsnet_amd.hip import conv as _conv
    monkeypatch.setattr(_conv, "FORCE_3X3", 5)
    dtype, Ci, Co, N, H, W, G = torch.bfloat16, 16, 16, 4, 40, 72, 2
    g = torch.Generator().manual_seed(911)
    x, gy = _bf(torch.randn(N, Ci, H, W, generator=g)), _bf(torch.randn(N, Co, H, W, generator=g))
    w = _bf(torch.randn(Co, Ci, 3, 3, generator=g) / 12.0)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dtype, dev)
    op.pack(w.to(dev).contiguous())
    assert op.plan_3x3(N, H, W, forward=True)["kernel"] == "p1"
    xd, gyd = _nhwc(x, dev, dtype), _nhwc(gy, dev, dtype)
    stats = torch.zeros(G, 8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(xd, stats=stats, stat_groups=G, out_f32=True)
    torch.cuda.synchronize()
    y_ref = F.conv2d(x, w, padding=1)
    ys = y_ref.abs().max().item()
    assert (y.permute(0, 3, 1, 2).cpu() - y_ref).abs().max().item() <= 2e-3 * ys
    s = stats.sum(1).cpu()
    for k in range(G):
        part = y_ref[k * (N // G):(k + 1) * (N // G)].double()
        assert torch.allclose(s[k, 0, :Co], part.sum(dim=(0, 2, 3)), rtol=1e-3, atol=1e-3 * ys * (N * H * W) ** 0.5)
        assert torch.allclose(s[k, 1, :Co], (part ** 2).sum(dim=(0, 2, 3)), rtol=2e-3)
    # data gradient with the derived sums of the BatchNorm in front, two groups
    yact, cprev = (_bf(torch.randn(N, Ci, H, W, generator=g)) for _ in range(2))
    st = ops.BnState(Ci, dev, G)
    mean, invstd = torch.randn(G, Ci, generator=g) * 0.1, torch.rand(G, Ci, generator=g) + 0.5
    st.mean.copy_(mean.reshape(st.mean.shape)); st.invstd.copy_(invstd.reshape(st.invstd.shape)); st.count = float(N // G * H * W)
    sums = torch.zeros(G * 8, 2, Ci, dtype=torch.float64, device=dev)
    d = op.dgrad(gyd, H, W, mask=_nhwc(yact, dev, dtype), bn_fuse=(_nhwc(cprev, dev, dtype), st, sums))
    torch.cuda.synchronize()
    dx_ref = F.conv_transpose2d(gy, w, padding=1)
    gref = torch.where(yact > 0, dx_ref, torch.zeros(()))
    assert (d.float().permute(0, 3, 1, 2).cpu() - gref).abs().max().item() <= 1e-2 * gref.abs().max().item()
    ss = sums.view(G, 8, 2, Ci).sum(1).cpu()
    for k in range(G):
        sl = slice(k * (N // G), (k + 1) * (N // G))
        xh = (cprev[sl] - mean[k].view(1, -1, 1, 1)) * invstd[k].view(1, -1, 1, 1)
        r0, r1 = gref[sl].double().sum(dim=(0, 2, 3)), (gref[sl].double() * xh.double()).sum(dim=(0, 2, 3))
        big = max(r0.abs().max().item(), r1.abs().max().item())
        eps = 2e-3 * big + 1e-3 * gref[sl].double().norm().item() / Ci ** 0.5
        assert (ss[k, 0] - r0).abs().max().item() <= eps and (ss[k, 1] - r1).abs().max().item() <= eps
