"""Two problems per launch (the fs_*2 entry points, include/fsnet_hip.h; EncoderPass in engine/nets.py): the depth
encoder's call and the stacked pose encoder's call of a training step (reference monodepth2_model.py:24-43:
self.depth_backbone(image) and self.pose_backbone(cat(pair)) once per source frame) share every post-stem launch.

Checked here: (1) value level — a two-weight launch against two torch conv2d calls / their autograd, per kernel family;
(2) a shared launch against the same two problems launched one after the other, bit for bit where no atomic is involved;
(3) the two-lane encoder pass against two one-lane passes; (4) the training step with and without lanes."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def to_nhwc(x, cp, dtype):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cp, dtype=dtype, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    return out


def _problem(dev, dtype, Ci, Co, k, stride, pad, N, H, W, seed):
    from fsnet_amd.hip.conv import ConvOp
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    gy = torch.randn(y_ref.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    y_ref.backward(gy)
    op = ConvOp(Ci, Co, k, k, stride, pad, dtype, dev, need_dgrad=True)
    op.pack(w.to(dev).contiguous())
    return dict(op=op, x=to_nhwc(x.to(dev), op.Ci_p, dtype), gy=to_nhwc(gy.to(dev), op.Co_p, dtype), y_ref=y_ref.detach(),
                dx_ref=xr.grad, dw_ref=wr.grad, Co=Co, Ci=Ci, H=H, W=W, k=k)


# (Ci, Co, k, stride, pad, H, W, N of lane 0, N of lane 1, statistics groups of lane 1)
LANE_CASES = [
    (64, 64, 3, 1, 1, 48, 160, 4, 8, 2),       # layer 1: the 32x32-tile kernel takes the pair (>= 512 tiles of 256 pixels)
    (64, 64, 3, 1, 1, 16, 32, 2, 4, 2),        # small: 16x16-tile kernel
    (128, 128, 3, 1, 1, 12, 20, 3, 6, 2),      # ragged tiles
    (64, 128, 3, 2, 1, 16, 32, 2, 4, 2),       # stage entry: stride-2 forward on the LDS-halo kernel, parity-class data gradient
    (64, 128, 1, 2, 0, 16, 32, 2, 4, 2),       # downsample projection: implicit GEMM both ways
    (256, 512, 3, 2, 1, 12, 40, 2, 4, 2),
    (512, 512, 3, 1, 1, 6, 20, 3, 6, 2),       # layer 4: groups of 360 rows (one statistics group per blockIdx.z)
    (6, 64, 7, 2, 3, 64, 128, 2, 4, 2),        # the stems pair although their real input channels differ (3 / 6): see below
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", LANE_CASES)
def test_two_weight_launch_against_two_conv2d_calls(dev, case, dtype):
    """forward (+ BatchNorm statistics per group), data gradient and weight gradient of TWO convolutions with different
    weights and batch sizes in one launch each == torch conv2d / autograd of each, and == the one-problem launches"""
    from fsnet_amd.hip.conv import run_specs
    Ci, Co, k, stride, pad, H, W, N0, N1, G1 = case
    P = [_problem(dev, dtype, 3 if (k == 7) else Ci, Co, k, stride, pad, N0, H, W, 11),
         _problem(dev, dtype, Ci, Co, k, stride, pad, N1, H, W, 12)]
    groups = [1, G1]
    tol = 2e-5 if dtype == torch.float32 else 2e-3

    def stats_for(p, G):
        shape = (8, 2, p["op"].Co_p) if G == 1 else (G, 8, 2, p["op"].Co_p)
        return torch.zeros(shape, dtype=torch.float64, device=dev)
    # ---- forward: shared launch, then one after the other
    st_pair = [stats_for(p, G) for p, G in zip(P, groups)]
    specs = [p["op"].forward_spec(p["x"], stats=s, stat_groups=G) for p, s, G in zip(P, st_pair, groups)]
    run_specs(specs)
    y_pair = [sp.out for sp in specs]
    st_solo = [stats_for(p, G) for p, G in zip(P, groups)]
    y_solo = [p["op"].forward(p["x"], stats=s, stat_groups=G) for p, s, G in zip(P, st_solo, groups)]
    torch.cuda.synchronize()
    import ctypes as C
    from fsnet_amd.hip.binding import lib
    same_kernel = True
    if k == 3 and stride == 1:
        # the pair's pixel-tile count can move it onto the other 3x3 kernel (32x32 MFMA tiles from 512 tiles of 256
        # pixels on): then the two results differ by the summation order inside a tile, not bit for bit
        plan = (C.c_int32 * 4)()
        kinds = []
        for sp in specs:
            assert lib.fs_conv3x3_halo_plan(C.byref(sp.a), sp.code, plan) == 0
            kinds.append(plan[0])
        assert lib.fs_conv3x3_halo2_plan(C.byref(specs[0].a), C.byref(specs[1].a), specs[0].code, plan) == 0
        same_kernel = all(kd == plan[0] for kd in kinds)
    for p, yp, ys, sp_, ss, G in zip(P, y_pair, y_solo, st_pair, st_solo, groups):
        if same_kernel:
            assert torch.equal(yp, ys)                               # same tiles, same arithmetic: bit for bit
        else:
            assert (yp.float() - ys.float()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 8e-3) * ys.float().abs().max().item()
        got = yp[..., :p["Co"]].permute(0, 3, 1, 2).float().cpu()
        assert (got - p["y_ref"]).abs().max().item() <= max(tol, 8e-3 if dtype == torch.bfloat16 else 0) * p["y_ref"].abs().max().item()
        # statistics per group: f64 atomics in a different order
        a, b = sp_.reshape(G, 8, 2, -1).sum(1), ss.reshape(G, 8, 2, -1).sum(1)
        st_tol = 1e-9 if same_kernel else (1e-5 if dtype == torch.float32 else 2e-3)
        assert torch.allclose(a, b, rtol=st_tol, atol=st_tol * float(b.abs().max()))
        n = p["y_ref"].shape[0] // G
        yq = yp.float() if dtype == torch.float32 else yp.float()
        for gi in range(G):
            ref = yq[gi * n:(gi + 1) * n].double()
            assert torch.allclose(a[gi, 0], ref.sum((0, 1, 2)), rtol=2e-3, atol=2e-3 * float(ref.abs().sum((0, 1, 2)).max()))
    # ---- data gradient
    if k != 7:
        dspecs = [p["op"].dgrad_spec(p["gy"], H, W) for p in P]
        run_specs(dspecs)
        dx_solo = [p["op"].dgrad(p["gy"], H, W) for p in P]
        torch.cuda.synchronize()
        for p, sp, ds in zip(P, dspecs, dx_solo):
            if same_kernel:
                assert torch.equal(sp.out, ds)
            got = sp.out[..., :p["Ci"]].permute(0, 3, 1, 2).float().cpu()
            assert (got - p["dx_ref"]).abs().max().item() <= (2e-5 if dtype == torch.float32 else 1e-2) * p["dx_ref"].abs().max().item()
    # ---- weight gradient: one launch, two dW
    dws = [torch.zeros(p["Co"], p["dw_ref"].shape[1], k, k, device=dev) for p in P]
    run_specs([p["op"].wgrad_spec(p["gy"], p["x"], dw) for p, dw in zip(P, dws)])
    torch.cuda.synchronize()
    for p, dw in zip(P, dws):
        scale = p["dw_ref"].abs().max().item()
        assert (dw.cpu() - p["dw_ref"]).abs().max().item() <= (5e-5 if dtype == torch.float32 else 2e-3) * scale
    # accumulation: a second shared launch doubles both
    run_specs([p["op"].wgrad_spec(p["gy"], p["x"], dw) for p, dw in zip(P, dws)])
    torch.cuda.synchronize()
    for p, dw in zip(P, dws):
        assert (dw.cpu() - 2 * p["dw_ref"]).abs().max().item() <= 2 * (5e-5 if dtype == torch.float32 else 2e-3) * p["dw_ref"].abs().max().item()


def test_pair_shares_one_launch(dev):
    """the plan of a pair: one launch whose grid holds both problems' blocks (the depth lane's blocks rounded up to whole
    XCD rounds); problems that do not agree on the instantiation report FS_EINVAL (they run as two launches)"""
    import ctypes as C
    from fsnet_amd.hip.binding import lib
    dtype = torch.bfloat16
    P = [_problem(dev, dtype, 64, 64, 3, 1, 1, 12, 48, 160, 1), _problem(dev, dtype, 64, 64, 3, 1, 1, 24, 48, 160, 2)]
    specs = [p["op"].forward_spec(p["x"]) for p in P]
    plan = (C.c_int32 * 4)()
    solo = []
    for sp in specs:
        assert lib.fs_conv3x3_halo_plan(C.byref(sp.a), sp.code, plan) == 0
        solo.append(list(plan))
    assert lib.fs_conv3x3_halo2_plan(C.byref(specs[0].a), C.byref(specs[1].a), specs[0].code, plan) == 0
    assert plan[0] == 1 and plan[2] == 256                       # 36 images: the 32x32-tile kernel, 256-pixel tiles
    assert solo[0][1] + solo[1][1] <= plan[1] <= solo[0][1] + solo[1][1] + 8 or solo[0][0] != plan[0]
    # a ReLU epilogue on one side only: no shared instantiation
    odd = P[1]["op"].forward_spec(P[1]["x"], relu=True)
    assert lib.fs_conv3x3_halo2_plan(C.byref(specs[0].a), C.byref(odd.a), specs[0].code, plan) == 1
    # ... and the launch entry then runs the two one after the other, same results
    from fsnet_amd.hip.conv import run_specs
    a = P[0]["op"].forward_spec(P[0]["x"])
    run_specs([a, odd])
    torch.cuda.synchronize()
    assert torch.equal(a.out, P[0]["op"].forward(P[0]["x"]))
    assert torch.equal(odd.out, P[1]["op"].forward(P[1]["x"], relu=True))
    # weight gradient: the pair's blocks fit the device's one round
    wp = (C.c_int32 * 4)()
    ws = [p["op"].wgrad_spec(p["gy"], p["x"], torch.zeros(64, 64, 3, 3, device=dev)) for p in P]
    assert lib.fs_conv_wgrad2_plan(C.byref(ws[0].a), C.byref(ws[1].a), ws[0].code, wp) == 0
    assert wp[0] == 1 and wp[1] <= wp[3], list(wp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_and_pool_pairs(dev, dtype):
    """fs_bn_apply2 / fs_bn_bwd_reduce2 / fs_bn_bwd_apply2 / fs_maxpool_*2: two tensors with their own BatchNorm modules
    and statistics groups in one launch == the one-problem launches, bit for bit"""
    from fsnet_amd.engine.nets import bn_tensors
    from fsnet_amd.hip import ops
    from fsnet_amd.hip.conv import run_specs
    Cc, H, W = 64, 12, 20
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(N, H, W, Cc, generator=g).to(dev).to(dtype) for N in (3, 6)]
    groups = [1, 2]
    bns = []
    for i in range(2):
        bn = torch.nn.BatchNorm2d(Cc).to(dev)
        bn.weight.data.uniform_(0.5, 1.5, generator=None); bn.bias.data.uniform_(-0.3, 0.3)
        bns.append(bn)

    def stats_of(x, G):
        n = x.shape[0] // G
        st = torch.zeros(G, 8, 2, Cc, dtype=torch.float64, device=dev)
        for gi in range(G):
            v = x[gi * n:(gi + 1) * n].double()
            st[gi, 0, 0] = v.sum((0, 1, 2)); st[gi, 0, 1] = (v * v).sum((0, 1, 2))
        return st if G > 1 else st[0]
    results = {}
    for mode in ("pair", "solo"):
        mods = [copy.deepcopy(b) for b in bns]
        sts = [ops.BnState(Cc, dev, G) for G in groups]
        ys = [torch.empty_like(x) for x in xs]
        stats = [stats_of(x, G) for x, G in zip(xs, groups)]     # (a Spec holds pointers, not tensors: keep them alive)
        specs = [ops.bn_apply_spec(x, s_, bn_tensors(m), st, y, H, W, (x.shape[0] // G) * H * W, relu=True, groups=G)
                 for x, s_, G, m, st, y in zip(xs, stats, groups, mods, sts, ys)]
        if mode == "pair":
            run_specs(specs)
        else:
            for sp in specs:
                run_specs([sp])
        # backward of relu(bn(x)) with an upstream gradient
        gg = torch.Generator().manual_seed(5)
        douts = [torch.randn(x.shape, generator=gg).to(dev).to(dtype) for x in xs]
        dxs = [torch.empty_like(x) for x in xs]
        gouts = [torch.empty_like(x) for x in xs]
        for m in mods:
            m.weight.grad = torch.zeros_like(m.weight); m.bias.grad = torch.zeros_like(m.bias)
        calls = [dict(dout=d, y=y, x=x, gamma=m.weight.data, st=st, dx=dx, dgamma=m.weight.grad, dbeta=m.bias.grad, H=H, W=W,
                      relu=True, g_out=go) for d, y, x, m, st, dx, go in zip(douts, ys, xs, mods, sts, dxs, gouts)]
        if mode == "pair":
            ops.bn_backward_multi(calls)
        else:
            for c in calls:
                ops.bn_backward_multi([c])
        pooled = ops.maxpool_fwd_multi(ys) if mode == "pair" else [ops.maxpool_fwd(y) for y in ys]
        up = [torch.randn(p[0].shape, generator=gg).to(dev).to(dtype) for p in pooled]
        adds = [None, torch.randn(ys[1].shape, generator=gg).to(dev).to(dtype)]
        if mode == "pair":
            back = ops.maxpool_bwd_multi(up, [p[1] for p in pooled], H, W, adds)
        else:
            back = [ops.maxpool_bwd(u, p[1], H, W, addend=a) for u, p, a in zip(up, pooled, adds)]
        torch.cuda.synchronize()
        results[mode] = dict(y=ys, dx=dxs, g=gouts, rm=[m.running_mean.clone() for m in mods],
                             nbt=[int(m.num_batches_tracked) for m in mods], dg=[m.weight.grad.clone() for m in mods],
                             db=[m.bias.grad.clone() for m in mods], pool=[p[0] for p in pooled], idx=[p[1] for p in pooled],
                             back=back, mean=[st.mean.clone() for st in sts])
    a, b = results["pair"], results["solo"]
    assert a["nbt"] == b["nbt"] == [1, 2]
    for key in ("y", "dx", "g", "rm", "pool", "idx", "back", "mean"):
        for u, v in zip(a[key], b[key]):
            assert torch.equal(u, v), key
    for key in ("dg", "db"):
        for u, v in zip(a[key], b[key]):
            assert torch.allclose(u, v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), key
    # and against torch: lane 1 is two BatchNorm calls in a row
    x1 = xs[1].float().permute(0, 3, 1, 2)
    ref_mod = copy.deepcopy(bns[1]).train()
    n = xs[1].shape[0] // 2
    ref = torch.cat([F.relu(ref_mod(x1[:n])), F.relu(ref_mod(x1[n:]))]).permute(0, 2, 3, 1)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert float((a["y"][1].float() - ref).abs().max()) < tol * float(ref.abs().max())
    assert float((a["rm"][1] - ref_mod.running_mean).abs().max()) < 1e-4


def _encoders(dev, dtype, norm_eval=False):
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.models.backbone.resnet import resnet
    RT.set_compute_dtype(dtype)
    nets = []
    for seed, nimg in ((0, 1), (1, 2)):
        torch.manual_seed(seed)
        m = resnet(18, pretrained=False, num_input_images=nimg, norm_eval=norm_eval)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.uniform_(-0.2, 0.2)
        nets.append(m.to(dev).train())
    return nets


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("hw,norm_eval", [((64, 128), False), ((96, 160), False), ((64, 128), True)])
def test_two_lane_pass_equals_two_passes(dev, dtype, tol, hw, norm_eval):
    """forward_lanes(depth, image, pose, pairs) == depth(image) and pose.forward_pairs(pairs): features, parameter
    gradients, running statistics of both networks (norm_eval: BatchNorms in eval mode inside the training step)"""
    from fsnet_amd.vision_base.networks.models.backbone.resnet import forward_lanes, lanes_compatible
    H, W = hw
    B = 3
    da, pa = _encoders(dev, dtype, norm_eval)
    db, pb = copy.deepcopy(da), copy.deepcopy(pa)
    assert lanes_compatible(da, pa)
    g = torch.Generator(device="cpu").manual_seed(1)
    imgs = [torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(3)]
    pairs = [(imgs[1], imgs[0]), (imgs[0], imgs[2])]
    fa = [da(imgs[0]), pa.forward_pairs(pairs)]
    fb = list(forward_lanes(db, imgs[0], pb, pairs))
    ups = []
    for lane in range(2):
        for i in range(5):
            a, b = fa[lane][i].float(), fb[lane][i].float()
            assert a.shape == b.shape
            err = float(((a - b).norm() / a.norm().clamp_min(1e-6)).detach())
            assert err < tol, (lane, i, err)
        # the depth decoder reads all five features, the pose decoder only the last one
        ups.append([torch.randn(f.shape, generator=g).to(dev).to(f.dtype) if (lane == 0 or i == 4) else None
                    for i, f in enumerate(fa[lane])])
    la = sum((f.float() * u.float()).sum() for lane in range(2) for f, u in zip(fa[lane], ups[lane]) if u is not None)
    lb = sum((f.float() * u.float()).sum() for lane in range(2) for f, u in zip(fb[lane], ups[lane]) if u is not None)
    la.backward()
    lb.backward()
    torch.cuda.synchronize()
    for ma, mb in ((da, db), (pa, pb)):
        for (n, p1), (_, p2) in zip(ma.named_parameters(), mb.named_parameters()):
            if p1.grad is None:
                assert p2.grad is None or float(p2.grad.abs().max()) == 0, n
                continue
            ga, gb = p1.grad.float(), p2.grad.float()
            rel = float((ga - gb).norm() / ga.norm().clamp_min(1e-9))
            assert rel < (1e-3 if dtype == torch.float32 else 8e-2), (n, rel)
        for (n, b1), (_, b2) in zip(ma.named_buffers(), mb.named_buffers()):
            if n.endswith("num_batches_tracked"):
                assert int(b1) == int(b2), n
            else:
                assert float((b1 - b2).abs().max()) < tol * float(b1.abs().max().clamp_min(1.0)), n


def test_lanes_need_matching_encoders(dev):
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.models.backbone.resnet import lanes_compatible, resnet
    RT.set_compute_dtype(torch.float32)
    a = resnet(18, pretrained=False, norm_eval=False).to(dev).train()
    b = resnet(34, pretrained=False, num_input_images=2, norm_eval=False).to(dev).train()
    c = resnet(18, pretrained=False, num_input_images=2, norm_eval=True).to(dev).train()
    d = resnet(18, pretrained=False, num_input_images=2, norm_eval=False, frozen_stages=1).to(dev).train()
    e = resnet(18, pretrained=False, num_input_images=2, norm_eval=False).to(dev).train()
    assert not lanes_compatible(a, b) and not lanes_compatible(a, c) and not lanes_compatible(a, d)
    assert lanes_compatible(a, e)
    assert not lanes_compatible(a, e.eval())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-3)])
def test_training_step_with_and_without_lanes(dev, dtype, tol):
    """MonoDepthMeta.forward_train through the two-lane pass == the two-chain step of rounds 1-4 (FSNET_AMD_LANES=0):
    loss and every parameter gradient"""
    from oracle import fsnet_oracle as O
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(dtype)
    RT.tie_noise = False
    res = {}
    try:
        for flag in (False, True):
            RT.lanes = flag
            m = build(**meta_arch_cfg(64, 128, with_pose=True))
            m.load_state_dict(O.init_state(seed=5, with_pose=True), strict=True)
            m = m.to(dev).train()
            data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(2, 64, 128, seed=7).items()}
            out = m(data, dict(is_training=True))
            out["loss"].backward()
            torch.cuda.synchronize()
            res[flag] = (float(out["loss"]), {n: p.grad.detach().float().clone() for n, p in m.named_parameters()},
                         {n: b.detach().clone() for n, b in m.named_buffers()})
    finally:
        RT.lanes = "auto"
        RT.tie_noise = True
        RT.set_compute_dtype(torch.bfloat16)
    assert abs(res[True][0] - res[False][0]) <= tol * abs(res[False][0])
    worst = 0.0
    gmax = max(float(g.norm()) for g in res[False][1].values())
    for n, ga in res[False][1].items():
        gb = res[True][1][n]
        if float(ga.norm()) < 1e-4 * gmax:
            # (convolution biases in front of a BatchNorm: the true gradient is zero, what is there is the rounding of
            # sums taken in another order)
            assert float(gb.norm()) < 1e-3 * gmax, n
            continue
        rel = float((ga - gb).norm() / ga.norm().clamp_min(1e-8))
        worst = max(worst, rel)
        assert rel < (2e-3 if dtype == torch.float32 else 0.15), (n, rel)
    for n, ba in res[False][2].items():
        bb = res[True][2][n]
        if n.endswith("num_batches_tracked"):
            assert int(ba) == int(bb), n
        elif ba.dtype.is_floating_point:
            assert float((ba.float() - bb.float()).abs().max()) <= 1e-3 * float(ba.float().abs().max().clamp_min(1.0)), n
