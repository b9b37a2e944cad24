"""GPU parity of the implicit-GEMM conv kernels (forward, dgrad, wgrad) against the CPU oracle
for this op, torch's fp32 conv2d + autograd (the ATen op the reference calls)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Ci, Co, k, stride, pad, N, H, W
    (3, 64, 7, 2, 3, 2, 32, 64),      # depth stem
    (6, 64, 7, 2, 3, 2, 32, 64),      # pose stem
    (64, 64, 3, 1, 1, 2, 16, 32),     # layer1
    (64, 128, 3, 2, 1, 2, 16, 32),    # stage entry
    (64, 128, 1, 2, 0, 2, 16, 32),    # downsample
    (128, 128, 3, 1, 1, 3, 12, 20),   # ragged M
    (256, 512, 3, 2, 1, 2, 12, 40),
    (512, 512, 3, 1, 1, 12, 6, 20),   # layer4 at bench batch
    (96, 32, 3, 1, 1, 2, 24, 40),     # decoder skip concat
    (32, 16, 3, 1, 1, 2, 24, 40),
    (16, 16, 3, 1, 1, 2, 48, 64),     # 16-channel halo path (half-filled chunk) + narrow wgrad, CIT = 16
    (32, 16, 3, 1, 1, 2, 48, 96),     # narrow wgrad, CIT = 32
    (64, 16, 3, 1, 1, 3, 40, 56),     # narrow wgrad over two input-channel tiles, ragged tiles
    (96, 32, 3, 1, 1, 2, 48, 96),     # narrow wgrad with 32 output channels (2 x 2 waves: channels x pixels)
    (512, 256, 1, 1, 0, 2, 6, 20),    # pose squeeze
    (256, 12, 1, 1, 0, 2, 6, 20),     # pose out (Co padded to 16)
    (64, 64, 3, 1, 1, 12, 48, 160),   # big M -> 128x64 tiles
    (128, 128, 3, 1, 1, 12, 48, 160), # big M -> 128x128 tiles
    (512, 512, 3, 1, 1, 4, 2, 3),     # layer4 of a 64x96 input: feature map narrower than the smallest tile
    (64, 64, 3, 1, 1, 3, 1, 2),
    # Bottleneck 1x1s on the row-streaming GEMM (conv1x1.hip; bf16) / implicit GEMM (fp32)
    (64, 256, 1, 1, 0, 2, 20, 24),    # expand, K = 64 (one 64-channel chunk)
    (256, 64, 1, 1, 0, 2, 20, 24),    # reduce, K = 256
    (128, 512, 1, 1, 0, 3, 9, 13),    # K = 128, ragged M (351 rows)
    (2048, 512, 1, 1, 0, 2, 5, 8),    # eight K chunks
    (256, 512, 1, 2, 0, 2, 20, 24),   # strided projection (downsample)
    (512, 96, 1, 1, 0, 2, 6, 10),     # Co_p = 96: 32-channel tiles
    # ... and on its LDS-staged GEMM form (conv1x1_gemm.hip) at sizes that reach every tile configuration
    (256, 1024, 1, 1, 0, 4, 20, 64),  # 128 x 128 tiles, eight K stages
    (64, 256, 1, 1, 0, 8, 80, 128),   # 256 x 128 tiles
    (256, 64, 1, 1, 0, 8, 80, 256),   # 256 x 64 tiles
    (1024, 512, 1, 2, 0, 2, 20, 64),  # strided projection, 32 K stages
    (128, 72, 1, 1, 0, 3, 17, 23),    # ragged M, Co_p = 80 (a partly filled channel tile)
    # ... whose weight gradients run on the LDS-DMA 1x1 kernel (wgrad1x1_kernel: >= 2048 pixels, 64-channel tiles)
    (64, 64, 1, 1, 0, 4, 40, 64),     # 64 x 64 tiles
    (128, 128, 1, 1, 0, 3, 27, 31),   # 2511 pixels: the last stage is ragged (rows past the end read as zero)
    (256, 512, 1, 2, 0, 4, 40, 64),   # strided projection with 2560 output pixels
]


def to_nhwc(x, cp, dtype):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cp, dtype=dtype, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_bwd(dev, case, dtype):
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, k, stride, pad, N, H, W = case
    g = torch.Generator().manual_seed(1234 + Ci * 7 + Co)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:  # compare like with like: oracle sees the bf16-rounded operands
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, stride=stride, padding=pad)
    gy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    y_ref.backward(gy)

    op = ConvOp(Ci, Co, k, k, stride, pad, dtype, dev, need_dgrad=True)
    op.pack(w.to(dev).contiguous())
    xd = to_nhwc(x.to(dev), op.Ci_p, dtype)
    bias = torch.zeros(op.Co_p, device=dev)
    bias[:Co] = b.to(dev)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(xd, bias=bias, stats=stats, out_f32=True)
    torch.cuda.synchronize()
    y_nchw = y[..., :Co].permute(0, 3, 1, 2).float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    scale = y_ref.abs().max().item()
    assert (y_nchw - y_ref.detach()).abs().max().item() <= tol * scale
    if Co % 16 != 0:
        assert y[..., Co:].abs().max().item() == 0
    # fused batch statistics
    s1 = y_ref.detach().double().sum(dim=(0, 2, 3))
    s2 = (y_ref.detach().double() ** 2).sum(dim=(0, 2, 3))
    stats = stats.sum(0)
    assert torch.allclose(stats[0, :Co].cpu(), s1, rtol=1e-3, atol=1e-3 * s2.max().sqrt().item())
    assert torch.allclose(stats[1, :Co].cpu(), s2, rtol=2e-3)

    # dgrad
    gyd = to_nhwc(gy.to(dev), op.Co_p, dtype)
    dx = op.dgrad(gyd, H, W)
    torch.cuda.synchronize()
    dx_nchw = dx[..., :Ci].permute(0, 3, 1, 2).float().cpu()
    gscale = xr.grad.abs().max().item()
    tol_g = 2e-5 if dtype == torch.float32 else 1e-2
    assert (dx_nchw - xr.grad).abs().max().item() <= tol_g * gscale

    # wgrad
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    op.wgrad(gyd, xd, dw)
    torch.cuda.synchronize()
    wscale = wr.grad.abs().max().item()
    tol_w = 5e-5 if dtype == torch.float32 else 2e-3
    assert (dw.cpu() - wr.grad).abs().max().item() <= tol_w * wscale


def test_conv_addend_relu_strided(dev):
    """epilogue addend + relu, strided (padded-buffer interior) source and destination."""
    from fsnet_amd.hip.conv import ConvOp
    dtype = torch.float32
    N, Ci, Co, H, W = 2, 32, 32, 10, 12
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 17
    add = torch.randn(N, Co, H, W, generator=g)
    ref = F.relu(F.conv2d(x, w, None, padding=1) + add)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dtype, dev)
    op.pack(w.to(dev))
    xp = torch.zeros(N, H + 2, W + 2, Ci, device=dev)
    xp[:, 1:-1, 1:-1] = x.to(dev).permute(0, 2, 3, 1)
    outp = torch.full((N, H + 2, W + 2, Co), -5.0, device=dev)
    addd = add.to(dev).permute(0, 2, 3, 1).contiguous()
    op.forward(xp[:, 1:-1, 1:-1], out=outp[:, 1:-1, 1:-1], addend=addd, relu=True)
    torch.cuda.synchronize()
    got = outp[:, 1:-1, 1:-1].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 1e-4
    assert (outp[:, 0] == -5).all() and (outp[:, :, 0] == -5).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(64, 64, 3, 1, 1, 4, 16, 32, 1), (64, 128, 3, 2, 1, 4, 16, 32, 1),
                                  (128, 128, 3, 1, 1, 4, 12, 20, 2), (256, 256, 3, 1, 1, 12, 6, 20, 1),
                                  (64, 128, 3, 2, 1, 4, 32, 64, 2), (64, 256, 1, 1, 0, 4, 16, 32, 1),
                                  (512, 128, 1, 1, 0, 4, 8, 16, 2)])
def test_dgrad_epilogue_carries_bn_backward_sums(dev, case, dtype):
    """ConvOp.dgrad(mask=, addend=, bn_fuse=): masked gradient and BatchNorm-backward sums (sum g, sum g*xhat)
    == unfused dgrad followed by fs_bn_bwd_reduce (batch_norm backward's first pass, ATen via
    vision_base/networks/models/backbone/resnet.py:17-41)."""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, k, stride, pad, N, H, W, G = case
    g = torch.Generator().manual_seed(7 + Ci + Co + stride)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    op = ConvOp(Ci, Co, k, k, stride, pad, dtype, dev, need_dgrad=True)
    op.pack(w.to(dev).contiguous())
    if not op.can_fuse_bn_bwd(N, H, W, G):
        pytest.skip("shape falls back to the unfused path by design")
    dy = torch.randn(N, Ho, Wo, Co, generator=g).to(dev).to(dtype)
    c = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)             # BN input (previous conv output)
    yact = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)          # its activation: the ReLU mask
    addend = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    st = ops.BnState(Ci, dev, G)
    st.mean.copy_(torch.randn(G * Ci, generator=g) * 0.1)
    st.invstd.copy_(torch.rand(G * Ci, generator=g) + 0.5)
    st.count = float(N // G * H * W)
    # unfused: dgrad (+addend), then the reduce pass with the ReLU mask
    d_ref = op.dgrad(dy, H, W, addend=addend)
    sums_ref = torch.zeros(G * 8, 2, Ci, dtype=torch.float64, device=dev)
    dx = torch.empty_like(c)
    gamma = torch.ones(Ci, device=dev)
    gout = torch.empty_like(c)
    ops.bn_backward(d_ref, yact, c, gamma, st, dx, None, None, H, W, relu=True, g_out=gout, sums=sums_ref, sums_zeroed=True)
    # fused
    sums = torch.zeros(G * 8, 2, Ci, dtype=torch.float64, device=dev)
    d_fused = op.dgrad(dy, H, W, addend=addend, mask=yact, bn_fuse=(c, st, sums))
    dx2 = torch.empty_like(c)
    ops.bn_backward(d_fused, None, c, gamma, st, dx2, None, None, H, W, sums=sums, reduced=True)
    torch.cuda.synchronize()
    assert torch.equal(d_fused, gout)                      # same masked gradient, bit for bit
    a = sums.view(G, 8, 2, Ci).sum(1).cpu()
    b = sums_ref.view(G, 8, 2, Ci).sum(1).cpu()
    # fused sums use the fp32 accumulator before the store rounding; the reduce pass re-reads the rounded tensor
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((a - b).abs().max()) <= tol * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())
    assert float((dx2.float() - dx.float()).abs().max()) <= (1e-4 if dtype == torch.float32 else 5e-2) * float(dx.float().abs().max())


@pytest.mark.parametrize("Ci,N,H,W,groups", [(3, 2, 32, 64, 1), (6, 4, 64, 96, 2), (3, 3, 38, 70, 1), (6, 12, 192, 640, 2)])
def test_stem_lds_kernel(dev, Ci, N, H, W, groups):
    """7x7/s2 stem through conv_stem.hip (LDS-resident weights, LDS im2col, persistent blocks): output and the
    per-group BatchNorm statistics against torch's conv2d on the same bf16 operands; ragged tiles and a batch that
    spans several persistent blocks included."""
    from fsnet_amd.hip.conv import ConvOp
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(31 + Ci + H)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, Ci, 7, 7, generator=g) / (Ci * 49) ** 0.5).bfloat16().float()
    op = ConvOp(Ci, 64, 7, 7, 2, 3, dtype, dev, need_dgrad=False)
    assert op.stem_lds
    op.pack(w.to(dev).contiguous())
    xd = to_nhwc(x.to(dev), op.Ci_p, dtype)
    stats = torch.zeros(groups, 8, 2, 64, dtype=torch.float64, device=dev)
    y = op.forward(xd, stats=stats, stat_groups=groups)
    torch.cuda.synchronize()
    ref = F.conv2d(x.to(dev), w.to(dev), None, stride=2, padding=3)            # fp32 on the same operands
    got = y.float().permute(0, 3, 1, 2)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 6e-3 * scale                      # bf16 output rounding
    n = N // groups
    for gi in range(groups):
        r = ref[gi * n:(gi + 1) * n].double()
        yb = got[gi * n:(gi + 1) * n].double()        # statistics are taken from the fp32 accumulators
        s = stats[gi].sum(0).cpu()
        assert torch.allclose(s[0], r.sum(dim=(0, 2, 3)).cpu(), rtol=2e-3, atol=2e-3 * (r ** 2).sum(dim=(0, 2, 3)).max().sqrt().item())
        assert torch.allclose(s[1], (r ** 2).sum(dim=(0, 2, 3)).cpu(), rtol=3e-3)
        assert yb.shape == r.shape
    # weight gradient (wgrad_stem_kernel from 4096 output pixels on, the generic kernel below) against autograd
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).bfloat16().float()
    xr, wr = x.clone().to(dev), w.clone().to(dev).requires_grad_(True)
    F.conv2d(xr, wr, None, stride=2, padding=3).backward(gy.to(dev))
    dw = torch.zeros(64, Ci, 7, 7, device=dev)
    op.wgrad(to_nhwc(gy.to(dev), 64, dtype), xd, dw)
    torch.cuda.synchronize()
    assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item()
    op.wgrad(to_nhwc(gy.to(dev), 64, dtype), xd, dw)           # accumulates into dw
    torch.cuda.synchronize()
    assert (dw - 2 * wr.grad).abs().max().item() <= 4e-3 * wr.grad.abs().max().item()
    # the generic path on the same inputs agrees to bf16 rounding
    import fsnet_amd.hip.conv as CV
    op2 = ConvOp(Ci, 64, 7, 7, 2, 3, dtype, dev, need_dgrad=False)
    op2.stem_lds = False
    op2.pack(w.to(dev).contiguous())
    y2 = op2.forward(xd)
    torch.cuda.synchronize()
    assert (y2.float() - y.float()).abs().max().item() <= 1.6e-2 * scale


@pytest.mark.gpu
def test_pack_table_follows_every_operand_buffer(dev):
    """the cached descriptor table of the one-launch weight re-pack is keyed by everything it encodes: a layer whose
    dgrad operand moved (a dropped model's successor at the same master-weight addresses) must get a fresh table —
    the stale one packed into freed memory and left the new operand empty (a rare wrong feature gradient)"""
    from fsnet_amd.engine.nets import ConvLayer
    from fsnet_amd.engine.runtime import RT
    conv = torch.nn.Conv2d(16, 32, 3, padding=1, bias=False).to(dev)
    cl = ConvLayer(conv)
    op = cl.ready(torch.bfloat16, dev)
    torch.cuda.synchronize()
    ref_d, ref_f = op.w_d.clone(), op.w_f.clone()
    assert float(ref_d.float().abs().sum()) > 0
    op.w_d = torch.zeros_like(op.w_d)
    RT.bump_weights()
    assert cl.ready(torch.bfloat16, dev) is op
    torch.cuda.synchronize()
    assert torch.equal(op.w_d, ref_d) and torch.equal(op.w_f, ref_f)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(64, 128, 2, 16, 32, 1), (64, 128, 3, 20, 44, 1), (128, 256, 4, 24, 80, 2),
                                  (256, 512, 6, 12, 40, 2), (64, 128, 2, 96, 160, 1)])
def test_stride2_data_gradient_on_the_class_fused_kernel(dev, case, dtype):
    """fs_conv3x3_s2d (conv3x3_s2d.hip: the four output-parity classes of a 3x3 / stride-2 data gradient from one staged
    dY halo) takes the ResNet stage entries' launches and equals autograd's convolution_backward(input) of
    resnet.py:33-50 with stride 2, with every epilogue the encoders use (residual addend, ReLU mask, BatchNorm-backward
    sums per statistics group) — and the implicit GEMM's class launch it replaces (the path of rounds 1-4)."""
    import ctypes as C
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.binding import lib, stream_ptr
    from fsnet_amd.hip.conv import ConvOp, LaunchProfile
    Ci, Co, N, H, W, G = case
    g = torch.Generator().manual_seed(3 + Ci + H)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    gy = torch.randn(N, Co, H // 2, W // 2, generator=g)
    if dtype == torch.bfloat16:
        x, w, gy = x.bfloat16().float(), w.bfloat16().float(), gy.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, stride=2, padding=1).backward(gy)
    op = ConvOp(Ci, Co, 3, 3, 2, 1, dtype, dev, need_dgrad=True)
    op.pack(w.to(dev).contiguous())
    gyd = to_nhwc(gy.to(dev), op.Co_p, dtype)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    # plain
    LaunchProfile.begin()
    dx = op.dgrad(gyd, H, W)
    kinds = [k for (k, _, _) in LaunchProfile.end()]
    assert kinds == ["conv3x3_s2d"], kinds                # the new kernel took the launch
    got = dx[..., :Ci].permute(0, 3, 1, 2).float().cpu()
    assert (got - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    # against the implicit GEMM's class launch, same arguments
    sp = op.dgrad_spec(gyd, H, W)
    assert lib.fs_conv_igemm(C.byref(sp.a), sp.code, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert (sp.out.float() - dx.float()).abs().max().item() <= (2e-5 if dtype == torch.float32 else 8e-3) * dx.float().abs().max().item()
    # addend + mask + BatchNorm-backward sums, G statistics groups
    c = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    yact = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    addend = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    st = ops.BnState(Ci, dev, G)
    st.mean.copy_(torch.randn(G * Ci, generator=g) * 0.1)
    st.invstd.copy_(torch.rand(G * Ci, generator=g) + 0.5)
    st.count = float(N // G * H * W)
    sums = torch.zeros(G * 8, 2, Ci, dtype=torch.float64, device=dev)
    LaunchProfile.begin()
    d_fused = op.dgrad(gyd, H, W, addend=addend, mask=yact, bn_fuse=(c, st, sums))
    kinds = [k for (k, _, _) in LaunchProfile.end()]
    assert kinds == ["conv3x3_s2d"], kinds
    ref = (dx.float() + addend.float()) * (yact.float() > 0)
    assert (d_fused.float() - ref).abs().max().item() <= (1e-5 if dtype == torch.float32 else 1.6e-2) * ref.abs().max().item()
    n = N // G
    gq = d_fused.float().double().view(G, n, H, W, Ci)
    xhat = (c.float().double().view(G, n, H, W, Ci) - st.mean.double().view(G, 1, 1, 1, Ci)) * st.invstd.double().view(G, 1, 1, 1, Ci)
    a = sums.view(G, 8, 2, Ci).sum(1)
    s_tol = 1e-5 if dtype == torch.float32 else 2e-2
    for k, refsum in enumerate((gq.sum((1, 2, 3)), (gq * xhat).sum((1, 2, 3)))):
        assert float((a[:, k] - refsum).abs().max()) <= s_tol * float(refsum.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(64, 128, 2, 16, 32, 1), (128, 256, 4, 24, 80, 2), (256, 512, 3, 12, 40, 1)])
def test_stride2_data_gradient_carries_the_downsample_projection(dev, case, dtype):
    """fs_conv3x3_s2d with FsConvArgs.ds_src: dgrad(conv1: 3x3/s2) + dgrad(downsample: 1x1/s2) of a stage-entry
    BasicBlock (resnet.py:33-50, 152-160: both read the block input) in one launch == autograd of the two convolutions,
    alone and with the block's epilogue (feature-gradient addend, ReLU mask, BatchNorm-backward sums)."""
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import ConvOp, LaunchProfile
    Ci, Co, N, H, W, G = case
    g = torch.Generator().manual_seed(5 + Ci + H)
    x = torch.randn(N, Ci, H, W, generator=g)
    w3 = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    w1 = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    gy3 = torch.randn(N, Co, H // 2, W // 2, generator=g)
    gy1 = torch.randn(N, Co, H // 2, W // 2, generator=g)
    if dtype == torch.bfloat16:
        x, w3, w1, gy3, gy1 = (t.bfloat16().float() for t in (x, w3, w1, gy3, gy1))
    xr = x.clone().requires_grad_(True)
    (F.conv2d(xr, w3, None, stride=2, padding=1) * gy3).sum().backward()
    (F.conv2d(xr, w1, None, stride=2, padding=0) * gy1).sum().backward()
    op3 = ConvOp(Ci, Co, 3, 3, 2, 1, dtype, dev, need_dgrad=True)
    op1 = ConvOp(Ci, Co, 1, 1, 2, 0, dtype, dev, need_dgrad=True)
    op3.pack(w3.to(dev).contiguous()); op1.pack(w1.to(dev).contiguous())
    d3, d1 = to_nhwc(gy3.to(dev), op3.Co_p, dtype), to_nhwc(gy1.to(dev), op1.Co_p, dtype)
    assert op3.can_fold_ds_dgrad(op1, d3, d1)
    LaunchProfile.begin()
    dx = op3.dgrad(d3, H, W, ds=(op1, d1))
    kinds = [k for (k, _, _) in LaunchProfile.end()]
    assert kinds == ["conv3x3_s2d"], kinds
    got = dx[..., :Ci].permute(0, 3, 1, 2).float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert (got - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    # == the two launches it replaces (the projection's result as the addend)
    two = op3.dgrad(d3, H, W, addend=op1.dgrad(d1, H, W))
    assert (two.float() - dx.float()).abs().max().item() <= (2e-5 if dtype == torch.float32 else 1.6e-2) * dx.float().abs().max().item()
    # with the block's epilogue
    c = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    yact = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    addend = torch.randn(N, H, W, Ci, generator=g).to(dev).to(dtype)
    st = ops.BnState(Ci, dev, G)
    st.mean.copy_(torch.randn(G * Ci, generator=g) * 0.1)
    st.invstd.copy_(torch.rand(G * Ci, generator=g) + 0.5)
    st.count = float(N // G * H * W)
    sums = torch.zeros(G * 8, 2, Ci, dtype=torch.float64, device=dev)
    d_fused = op3.dgrad(d3, H, W, addend=addend, mask=yact, bn_fuse=(c, st, sums), ds=(op1, d1))
    torch.cuda.synchronize()
    ref = (dx.float() + addend.float()) * (yact.float() > 0)
    assert (d_fused.float() - ref).abs().max().item() <= (1e-5 if dtype == torch.float32 else 1.6e-2) * ref.abs().max().item()
    n = N // G
    gq = d_fused.float().double().view(G, n, H, W, Ci)
    xhat = (c.float().double().view(G, n, H, W, Ci) - st.mean.double().view(G, 1, 1, 1, Ci)) * st.invstd.double().view(G, 1, 1, 1, Ci)
    a = sums.view(G, 8, 2, Ci).sum(1)
    s_tol = 1e-5 if dtype == torch.float32 else 2e-2
    for k, refsum in enumerate((gq.sum((1, 2, 3)), (gq * xhat).sum((1, 2, 3)))):
        assert float((a[:, k] - refsum).abs().max()) <= s_tol * float(refsum.abs().max())


@pytest.mark.parametrize("case", [(64, 256, 4, 40, 64, 2, False), (256, 128, 4, 40, 64, 2, True), (512, 128, 6, 9, 16, 3, False),
                                  (128, 64, 8, 80, 256, 2, True)])
def test_conv1x1_gemm_epilogues(dev, case):
    """fs_conv1x1's LDS-staged GEMM (conv1x1_gemm.hip): addend + ReLU, BatchNorm statistics per statistics group
    (stacked pose pairs), bf16 and fp32 stores, strided (channel-slice) source and destination views — against conv2d
    (resnet.py:52-89: conv1x1 -> bn -> relu of the Bottleneck)."""
    from fsnet_amd.hip.conv import ConvOp
    Ci, Co, N, H, W, G, f32 = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11 + Ci + Co)
    x = torch.randn(N, H, W, Ci, generator=g).to(dt)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).to(dt).float()
    add = torch.randn(N, H, W, Co, generator=g).to(dt)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
    op = ConvOp(Ci, Co, 1, 1, 1, 0, dt, dev, need_dgrad=True)
    op.pack(w.to(dev))
    # source and destination are channel slices of wider buffers (a concatenated skip tensor): pixel rows 2x apart
    xw = torch.zeros(N, H, W, 2 * Ci, dtype=dt, device=dev)
    xw[..., Ci:] = x.to(dev)
    yw = torch.full((N, H, W, 2 * Co), -3.0, dtype=torch.float32 if f32 else dt, device=dev)
    stats = torch.zeros(G, 8, 2, Co, dtype=torch.float64, device=dev)
    op.forward(xw[..., Ci:], out=yw[..., :Co], stats=stats, stat_groups=G, out_f32=f32)
    torch.cuda.synchronize()
    got = yw[..., :Co].float().cpu()
    assert (yw[..., Co:] == -3).all()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= (2e-3 if f32 else 1e-2) * scale
    n = N // G
    for k in range(G):
        r = ref[k * n:(k + 1) * n].double()
        s = stats[k].sum(0).cpu()
        assert torch.allclose(s[0], r.sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-3 * float((r ** 2).sum(dim=(0, 1, 2)).max().sqrt()))
        assert torch.allclose(s[1], (r ** 2).sum(dim=(0, 1, 2)), rtol=2e-3)
    # addend + ReLU (the Bottleneck's residual exit has no statistics: its BatchNorm sits in front of the add)
    y2 = op.forward(x.to(dev), addend=add.to(dev), relu=True)
    torch.cuda.synchronize()
    ref2 = F.relu(ref + add.float())
    assert float((y2.float().cpu() - ref2).abs().max()) <= 1e-2 * float(ref2.abs().max())
