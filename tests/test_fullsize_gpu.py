"""Properties that do not need an oracle, checked at the FULL benchmark size (BASELINE configs[1]: batch 12,
192x640, ResNet-18 depth+pose, bf16) where the CPU oracle would take minutes per step:
  * convolution forward is linear in its input, the weight gradient is additive over a split of the batch
    (fp32 accumulation: exact up to summation order) — on the real layer shapes;
  * one stacked pose pass == two separate calls (BatchNorm statistics groups) at full size;
  * identical source and target frames under the identity pose reproject onto themselves: zero photometric loss
    and zero pose gradient;
  * a few full-size training steps stay finite, replay from the hipGraph, and keep every parameter finite."""
import numpy as np
import pytest
import torch

from oracle import fsnet_oracle as O

pytestmark = pytest.mark.gpu
B, H, W = 12, 192, 640


@pytest.mark.parametrize("Ci,Co,h,w", [(64, 64, 48, 160), (512, 512, 6, 20), (16, 16, 192, 640)])
def test_conv_linearity_and_wgrad_additivity_on_benchmark_shapes(dev, Ci, Co, h, w):
    from fsnet_amd.hip.conv import ConvOp
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(Ci + h)
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    op.pack((torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(dev))
    # operands on a coarse binary grid: their sum is exactly representable in bf16, so linearity must hold to fp32
    # accumulation order
    x1 = (torch.randint(-8, 9, (B, h, w, op.Ci_p), generator=g).float() / 8).to(dev).to(dt)
    x2 = (torch.randint(-8, 9, (B, h, w, op.Ci_p), generator=g).float() / 8).to(dev).to(dt)
    y1 = op.forward(x1, out_f32=True)
    y2 = op.forward(x2, out_f32=True)
    y12 = op.forward(x1 + x2, out_f32=True)
    scale = float(y12.abs().max())
    assert float((y12 - (y1 + y2)).abs().max()) <= 2e-5 * scale
    dy = (torch.randint(-4, 5, (B, h, w, op.Co_p), generator=g).float() / 4).to(dev).to(dt)
    full = torch.zeros(Co, Ci, 3, 3, device=dev)
    op.wgrad(dy, x1, full)
    parts = torch.zeros(Co, Ci, 3, 3, device=dev)
    for lo, hi in ((0, 5), (5, B)):
        op.wgrad(dy[lo:hi].contiguous(), x1[lo:hi].contiguous(), parts)
    torch.cuda.synchronize()
    assert float((full - parts).abs().max()) <= 1e-4 * float(full.abs().max())
    # data gradient: adjoint identity <conv(x), dy> == <x, dgrad(dy)>
    lhs = float((y1.double() * dy.double()).sum())
    dx = op.dgrad(dy, h, w)
    rhs = float((x1.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 2e-2 * (abs(lhs) + float(y1.abs().max()))      # dx is stored in bf16


def test_stacked_pose_pairs_equal_separate_calls_at_full_size(dev):
    import copy
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.models.backbone.resnet import resnet
    RT.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    ma = resnet(18, pretrained=False, num_input_images=2, norm_eval=False).to(dev).train()
    mb = copy.deepcopy(ma)
    g = torch.Generator().manual_seed(2)
    imgs = [torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(3)]
    pairs = [(imgs[1], imgs[0]), (imgs[0], imgs[2])]
    with torch.no_grad():
        fa = [ma.forward_pair(*p) for p in pairs]
        fb = mb.forward_pairs(pairs)
    for k in range(2):
        for i in range(5):
            a, b = fa[k][i].float(), fb[i][k * B:(k + 1) * B].float()
            assert float((a - b).norm() / a.norm()) < 2e-4, (k, i)
    for (n, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        if "running" in n:
            assert float((ba - bb).abs().max()) < 1e-4 * float(ba.abs().max().clamp_min(1.0)), n


def test_identical_frames_reproject_onto_themselves(dev):
    """reference monodepth2_decoder.py:68-128: sampling the source at its own pixel centres returns the source, so
    with source == target and T = I the reprojection term vanishes for every depth"""
    from fsnet_amd.hip import ops
    S = 4
    pl = ops.PhotometricLoss(B, H, W, [0, 1, 2, 3], dev, 0.5, 100.0)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    P2 = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0]]).repeat(B, 1, 1).to(dev)
    T = torch.eye(4).repeat(B, 1, 1).to(dev)
    depths = [(torch.rand(B, 1, H >> s, W >> s, generator=g) * 20 + 2).to(dev) for s in range(S)]
    disps = [1.0 / d for d in depths]
    out = pl.forward(img, [img.clone(), img.clone()], P2, [T, T.clone()], None, depths, disps, noise_seed=-1)
    torch.cuda.synchronize()
    photo = out[:S].cpu()
    smooth = out[S:2 * S].cpu()
    # what is left per scale is the smoothness term only
    assert float((photo - smooth).abs().max()) < 1e-6, (photo, smooth)
    d_depth, d_disp, dT = pl.backward(None)
    torch.cuda.synchronize()
    assert float(dT[0].abs().max()) < 1e-6 and float(dT[1].abs().max()) < 1e-6


def test_full_size_training_steps(dev):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.bfloat16)
    RT.tie_noise = True
    torch.manual_seed(0)
    m = build(**meta_arch_cfg(H, W, with_pose=True)).to(dev).train()
    tc = training_cfg(clip_gradients=35.0, lr=1e-4)
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(graph_warmup=2, **tc.training_hook)
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(B, H, W, seed=1).items()}
    losses = []
    for it in range(8):                      # the same batch: the loss must go down
        out = hook(dict(batch), m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert hook.graph_captures == 1 and hook.graph_replays == 5
    assert all(l == l and 0 < l < 10 for l in losses), losses
    assert losses[-1] < losses[0], losses
    flat = torch.cat([p.detach().flatten() for p in m.parameters()])
    assert bool(torch.isfinite(flat).all())
    assert float(opt.grad_norm()) > 0
    RT.tie_noise = False


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] / [4]: ResNet-50 (Bottleneck) at 320x1024
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Ci,Co,k,h,w", [(256, 64, 1, 80, 256), (64, 256, 1, 80, 256), (1024, 2048, 1, 10, 32),
                                         (512, 512, 3, 10, 32), (2048, 256, 3, 10, 32), (1280, 256, 3, 20, 64)])
def test_r50_conv_shapes_linearity_and_adjoint_at_320x1024(dev, Ci, Co, k, h, w):
    """the Bottleneck's 1x1 convolutions with up to 2048 channels and the R50 decoder's wide skip-concat convolutions
     at their real sizes, batch 4: linearity, weight-gradient additivity, adjoint identity"""
    from fsnet_amd.hip.conv import ConvOp
    dt, Bs = torch.bfloat16, 4
    g = torch.Generator().manual_seed(Ci + Co + h)
    op = ConvOp(Ci, Co, k, k, 1, k // 2, dt, dev)
    op.pack((torch.randn(Co, Ci, k, k, generator=g) / (k * Ci ** 0.5)).to(dev))
    x1 = (torch.randint(-8, 9, (Bs, h, w, op.Ci_p), generator=g).float() / 8).to(dev).to(dt)
    x2 = (torch.randint(-8, 9, (Bs, h, w, op.Ci_p), generator=g).float() / 8).to(dev).to(dt)
    y1, y2, y12 = op.forward(x1, out_f32=True), op.forward(x2, out_f32=True), op.forward(x1 + x2, out_f32=True)
    assert float((y12 - (y1 + y2)).abs().max()) <= 2e-5 * float(y12.abs().max())
    dy = (torch.randint(-4, 5, (Bs, h, w, op.Co_p), generator=g).float() / 4).to(dev).to(dt)
    full, parts = torch.zeros(Co, Ci, k, k, device=dev), torch.zeros(Co, Ci, k, k, device=dev)
    op.wgrad(dy, x1, full)
    for lo, hi in ((0, 1), (1, Bs)):
        op.wgrad(dy[lo:hi].contiguous(), x1[lo:hi].contiguous(), parts)
    torch.cuda.synchronize()
    assert float((full - parts).abs().max()) <= 1e-4 * float(full.abs().max())
    lhs = float((y1.double() * dy.double()).sum())
    dx = op.dgrad(dy, h, w).double()
    rhs = float((x1.double() * dx).sum())
    # dx is stored in bf16: every term x * dx carries an independent relative rounding error of 2^-9, so the sum's
    # error is a random walk of size 2^-9 sqrt(sum (x dx)^2) — six standard deviations
    sigma = 2.0 ** -9 * float(((x1.double() * dx) ** 2).sum().sqrt())
    assert abs(lhs - rhs) <= 6 * sigma + 1e-6 * abs(lhs), (lhs, rhs, sigma)


@pytest.mark.parametrize("with_pose,base_fx,bins", [(True, None, 16), (False, 492.0, 64)],
                         ids=["configs2-depth+pose", "configs4-wpose-64bins-basefx"])
def test_r50_full_size_training_steps_320x1024(dev, with_pose, base_fx, bins):
    """ResNet-50 at 320x1024, bf16, batch 4: eight steps through the hook (hipGraph replay included) stay finite and
    bounded (a randomly initialised ResNet-50 does not improve monotonically in eight clipped steps: no claim about the
    loss going down), gradients flow, every parameter stays finite"""
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    Hh, Ww, Bs = 320, 1024, 4
    RT.set_compute_dtype(torch.bfloat16)
    RT.tie_noise = True
    torch.manual_seed(0)
    m = build(**meta_arch_cfg(Hh, Ww, with_pose=with_pose, depth=50, num_output_channels=bins, base_fx=base_fx)).to(dev).train()
    tc = training_cfg(clip_gradients=1.0, lr=1e-4)          # configs/multi_dataset_example clips at 1.0
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(graph_warmup=2, **tc.training_hook)
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(Bs, Hh, Ww, seed=2).items()}
    losses = []
    for it in range(8):
        out = hook(dict(batch), m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert hook.graph_captures == 1 and hook.graph_replays == 5
    assert all(l == l and 0 < l < 10 for l in losses), losses
    assert max(losses) < 1.5 * losses[0], losses
    assert float(opt.grad_norm()) > 0
    assert bool(torch.isfinite(torch.cat([p.detach().flatten() for p in m.parameters()])).all())
    RT.tie_noise = False


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: KITTI-360 fisheye at the reference's size (configs/kitti360_fisheye_example:72,83-86,198-207:
# ResNet-18 + FishEyeDecoder, 64 bins, 384x384, batch 16, max depth 150, weight decay 1e-5)
# ---------------------------------------------------------------------------------------------------------------
def _fisheye_batches(n, Bf, Hf, Wf, dev, seed):
    from fsnet_amd.vision_base.data.datasets.dataset_utils import collate_fn
    from fsnet_amd.vision_base.data.datasets.synthetic import SyntheticTripletDataset
    ds = SyntheticTripletDataset(size=n * Bf, height=Hf, width=Wf, seed=seed, fisheye=True)
    out = []
    for i in range(n):
        d = collate_fn([ds[i * Bf + j] for j in range(Bf)])
        out.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()})
    return out


def test_fisheye_identical_frames_reproject_onto_themselves_at_384(dev):
    """size-independent property of the Mei chain at full size: ray table x norm -> identity pose -> cam2image lands on
    the pixel it started from, so with source == target the photometric term and the pose gradient vanish for every
    depth (monodepth2_decoder.py:355-411, mei_fisheye_utils.py:14-187) — for both calibrations of the batch"""
    from fsnet_amd.hip import ops
    from fsnet_amd.monodepth.networks.utils.mei_fisheye_utils import MeiCameraProjection
    from fsnet_amd.vision_base.data.datasets.synthetic import synthetic_mei_calib
    Bf, Hf, Wf, S = 4, 384, 384, 4
    pl = ops.PhotometricLoss(Bf, Hf, Wf, [0, 1, 2, 3], dev, 0.5, 150.0)
    g = torch.Generator().manual_seed(5)
    img = torch.rand(Bf, 3, Hf, Wf, generator=g).to(dev)
    Ps, calibs = zip(*[synthetic_mei_calib(Hf, Wf, b % 2) for b in range(Bf)])
    P2 = torch.from_numpy(np.stack(Ps, 0))
    tabs, rows = MeiCameraProjection().tables(Hf, Wf, P2, list(calibs), dev)
    pl.stage_fisheye(tabs, rows)
    P2 = P2.to(dev)
    T = torch.eye(4).repeat(Bf, 1, 1).to(dev)
    norms = [(torch.rand(Bf, 1, Hf >> s, Wf >> s, generator=g) * 20 + 2).to(dev) for s in range(S)]
    disps = [1.0 / d for d in norms]
    out = pl.forward(img, [img.clone(), img.clone()], P2, [T, T.clone()], None, norms, disps, noise_seed=-1)
    torch.cuda.synchronize()
    photo, smooth = out[:S].cpu(), out[S:2 * S].cpu()
    assert float((photo - smooth).abs().max()) < 2e-5, (photo, smooth)
    d_depth, d_disp, dT = pl.backward(None)
    torch.cuda.synchronize()
    assert float(dT[0].abs().max()) < 1e-5 and float(dT[1].abs().max()) < 1e-5


def test_fisheye_full_size_training_steps_384(dev):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.bfloat16)
    RT.tie_noise = True
    torch.manual_seed(0)
    Bf, Hf, Wf = 16, 384, 384
    m = build(**meta_arch_cfg(Hf, Wf, with_pose=False, num_output_channels=64, max_depth=150.0, fisheye=True)).to(dev).train()
    tc = training_cfg(clip_gradients=35.0, lr=1e-4, weight_decay=1e-5)
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(graph_warmup=2, **tc.training_hook)
    batches = _fisheye_batches(2, Bf, Hf, Wf, dev, seed=9)
    batches[1]["calib_meta"] = batches[1]["calib_meta"][1:] + batches[1]["calib_meta"][:1]    # the cameras change places:
    batches[1]["P2"] = torch.cat([batches[1]["P2"][1:], batches[1]["P2"][:1]], 0)             # staged per step, also on replay
    losses = []
    for it in range(8):
        out = hook(dict(batches[it % 2]), m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert hook.graph_captures == 1 and hook.graph_replays == 5
    assert all(l == l and 0 < l < 10 for l in losses), losses
    assert min(losses[-2:]) < max(losses[:2]), losses
    flat = torch.cat([p.detach().flatten() for p in m.parameters()])
    assert bool(torch.isfinite(flat).all())
    RT.tie_noise = False
