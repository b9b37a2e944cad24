"""Generate golden vectors by running the REAL reference (/root/reference) on CPU.

Dev-only: runs in the build container (the reference never travels to the GPU box).  Imports the
reference with the import shims of SURVEY Appendix A, feeds seeded synthetic inputs and writes
inputs + reference outputs to tests/golden/*.npz.  It also cross-checks oracle/fsnet_oracle.py
against the reference while doing so (prints max deviations).

    python tools/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools", "ref_shims"), "/root/reference", ROOT]
tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = type("SummaryWriter", (), {})
sys.modules["torch.utils.tensorboard"] = tb
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.cuda.synchronize = lambda *a, **k: None

from easydict import EasyDict  # noqa: E402
from vision_base.utils.builder import build  # noqa: E402
from monodepth.networks.utils import monodepth_utils as mu  # noqa: E402
from oracle import fsnet_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def ref_model(H, W, with_pose=True, depth=18, frozen_stages=-1, norm_eval=False):
    enc = [64, 64, 128, 256, 512]
    bb = dict(name='vision_base.networks.models.backbone.resnet.resnet', depth=depth, pretrained=False,
              frozen_stages=frozen_stages, num_stages=4, out_indices=(-1, 0, 1, 2, 3), norm_eval=norm_eval,
              dilations=(1, 1, 1, 1))
    head = dict(name='monodepth.networks.models.heads.monodepth2_decoder.MonoDepth2Decoder', scales=[0, 1, 2, 3],
                height=H, width=W, min_depth=0.5, max_depth=100.0, overlapped_mask=True, is_log_image=False,
                depth_decoder_cfg=dict(name='monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder',
                                       num_ch_enc=np.array(enc), num_output_channels=16, use_skips=True,
                                       scales=[0, 1, 2, 3], min_depth=0.5, max_depth=100))
    kw = dict(depth_backbone_cfg=bb, head_cfg=head, train_cfg=EasyDict(frame_ids=[0, 1, -1]), test_cfg=EasyDict())
    if with_pose:
        head['pose_decoder_cfg'] = dict(name='monodepth.networks.models.heads.pose_decoder.PoseDecoder',
                                        num_ch_enc=np.array(enc), num_input_features=1,
                                        num_frames_to_predict_for=2, stride=1)
        kw['pose_backbone_cfg'] = dict(bb, num_input_images=2)
        name = 'monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthMeta'
    else:
        name = 'monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthWPose'
    return build(name=name, **kw)


def dev(a, b):
    return float((a.double() - b.double()).abs().max())


def gen_ops():
    """op-level vectors (tiny shapes) straight from the reference's functions."""
    g = torch.Generator().manual_seed(11)
    out = {}
    B, H, W = 2, 24, 40
    # --- geometry: BackprojectDepth + Project3D (monodepth_utils.py:101-165) ---
    depth = 0.5 + 30 * torch.rand(B, 1, H, W, generator=g)
    data = O.synthetic_batch(B, H, W, seed=5)
    K = np.zeros([B, 4, 4]); K[:, :3, :3] = data['P2'][:, :3, :3].numpy(); K[:, 3, 3] = 1
    invK = np.linalg.pinv(K)
    aa = 0.02 * torch.randn(B, 1, 3, generator=g)
    tr = 0.3 * torch.randn(B, 1, 3, generator=g)
    T = mu.transformation_from_parameters(aa, tr, invert=False)
    Ti = mu.transformation_from_parameters(aa, tr, invert=True)
    cam = mu.BackprojectDepth()(depth, torch.from_numpy(invK).float())
    pix = mu.Project3D()(cam, torch.from_numpy(K).float(), T, B, H, W)
    out.update(geo_depth=npy(depth), geo_P2=npy(data['P2']), geo_aa=npy(aa), geo_tr=npy(tr), geo_T=npy(T),
               geo_Tinv=npy(Ti), geo_pix=npy(pix))
    Ko, invKo = O.intrinsics(data['P2'])
    print("oracle pix dev", dev(O.project(O.backproject(depth, invKo), Ko, T, H, W), pix),
          "T dev", dev(O.transformation_from_parameters(aa, tr), T),
          dev(O.transformation_from_parameters(aa, tr, True), Ti))
    # --- SSIM / reprojection / smoothness (monodepth_utils.py:168-215; decoder.py:118-128) ---
    x = torch.rand(B, 3, H, W, generator=g); y = (x + 0.1 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1)
    s = mu.SSIM()(x, y)
    dec = ref_model(64, 128, True).head
    rl = dec.compute_reprojection_loss(x, y)
    disp = torch.rand(B, 1, H, W, generator=g)
    sm = mu.get_smooth_loss(disp, x)
    out.update(ssim_x=npy(x), ssim_y=npy(y), ssim_out=npy(s), reproj_out=npy(rl), smooth_disp=npy(disp),
               smooth_out=npy(sm))
    print("oracle ssim dev", dev(O.ssim(x, y), s), "reproj", dev(O.reprojection_loss(x, y), rl), "smooth",
          dev(O.smooth_loss(disp, x), sm))
    # --- depth head (depth_encoder.py:76-88,114-121) ---
    logits = 6 * torch.randn(B, 16, 6, 10, generator=g)
    dd = dec.depth_decoder
    d_ref = dd._gather_activation(logits)
    disp_ref = mu.depth_to_disp(d_ref, 0.5, 100.0)
    out.update(head_logits=npy(logits), head_bins=npy(dd.depth_bins), head_depth=npy(d_ref), head_disp=npy(disp_ref))
    print("oracle bins dev", dev(O.depth_bins(0.5, 100, 16), dd.depth_bins), "head",
          dev(O.gather_activation(logits, dd.depth_bins), d_ref))
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)


def gen_loss_chain():
    """MonoDepth2Decoder.loss on synthetic network outputs (no network): loss_dict, dL/d depth_s,
    dL/d cam_T_cam (monodepth2_decoder.py:61-116, 205-347)."""
    B, H, W = 2, 64, 96
    data = O.synthetic_batch(B, H, W, seed=21)
    data['patched_mask'][:, :6, :] = 0  # exercise the overlapped-mask / patched-mask path
    g = torch.Generator().manual_seed(3)
    dec = ref_model(H, W, True).head
    outputs = {}
    leaves = {}
    for s in range(4):
        h, w = H >> s, W >> s
        ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
        d = (4 + 25 * (1 - ys) + 3 * torch.rand(B, 1, h, w, generator=g)).requires_grad_(True)
        leaves[("depth", s)] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = mu.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        aa = (0.01 * torch.randn(B, 1, 3, generator=g)).requires_grad_(True)
        tr = torch.tensor([[[0.02, -0.01, -0.6 if f > 0 else 0.6]]]).repeat(B, 1, 1) + 0.02 * torch.randn(B, 1, 3, generator=g)
        tr.requires_grad_(True)
        leaves[("aa", f)], leaves[("tr", f)] = aa, tr
        outputs[("cam_T_cam", f)] = mu.transformation_from_parameters(aa, tr, invert=(f < 0))
    torch.manual_seed(0)
    res = dec.loss(outputs, data)
    res['loss'].backward()
    out = {"H": H, "W": W, "seed": 21, "total_loss": npy(res['loss'])}
    for k, v in res['loss_dict'].items():
        out["ld_" + k.replace('/', '_')] = npy(v)
    for s in range(4):
        out["depth_%d" % s] = npy(leaves[("depth", s)])
        out["gdepth_%d" % s] = npy(leaves[("depth", s)].grad)
        out["minidx_%d" % s] = 0
    for f in (1, -1):
        tag = "p" if f > 0 else "m"
        out["aa_" + tag], out["tr_" + tag] = npy(leaves[("aa", f)]), npy(leaves[("tr", f)])
        out["gaa_" + tag], out["gtr_" + tag] = npy(leaves[("aa", f)].grad), npy(leaves[("tr", f)].grad)
        out["warp0_" + tag] = npy(outputs[("original_image", f, 0)])[:, :, ::4, ::4]
        out["ovmask0_" + tag] = npy(outputs[("overlapped_mask", f, 0)])
    for f in (0, 1, -1):
        out["img_%s" % {0: "0", 1: "p", -1: "m"}[f]] = npy(data[("original_image", f)])
    out["P2"] = npy(data["P2"]); out["patched_mask"] = npy(data["patched_mask"])
    # cross-check the oracle
    o2 = {}
    lv = {}
    for s in range(4):
        d = leaves[("depth", s)].detach().clone().requires_grad_(True); lv[s] = d
        o2[("depth", s, s)] = d; o2[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        o2[("cam_T_cam", f)] = outputs[("cam_T_cam", f)].detach()
    tot, ld = O.photometric_loss(o2, data)
    tot.backward()
    print("oracle chain: loss dev", dev(tot, res['loss']), "gdepth0 dev rel",
          dev(lv[0].grad, leaves[("depth", 0)].grad) / float(leaves[("depth", 0)].grad.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "loss_chain.npz"), **out)


def gen_model(with_pose, tag, steps=3):
    """Full model: forward outputs, loss_dict, per-parameter grad norms, and parameter checksums
    after Adam steps driven by the reference's own BaseTrainingHook (clip 35.0)."""
    from vision_base.pipeline_hooks.train_val_hooks.base_training_hooks import BaseTrainingHook
    B, H, W = 2, 64, 128
    sd0 = O.init_state(seed=1, with_pose=with_pose)
    m = ref_model(H, W, with_pose)
    missing = m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m.train()
    names = [k for k, _ in m.named_parameters()]
    assert names == [k for k in sd0 if O.is_param(k)], "parameter order/name mismatch"
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    hook = BaseTrainingHook(clip_gradients=35.0)
    out = {"B": B, "H": H, "W": W, "init_seed": 1}
    tr = O.OracleTrainer(sd0, with_pose=with_pose)
    for it in range(steps):
        data = O.synthetic_batch(B, H, W, seed=100 + it)
        if it == 0:
            torch.manual_seed(0)
            res = m(dict(data), dict(is_training=True))
            # forward-only capture on a copy of BN state: redo state restore below
            m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
            feats = None
        # capture forward outputs of step `it` through a manual pass identical to the hook
        opt.zero_grad()
        torch.manual_seed(0)
        d2 = dict(data)
        res = m(d2, dict(is_training=True))
        res['loss'].mean().backward()
        gn = torch.stack([p.grad.norm() for p in m.parameters()])
        total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 35.0)
        opt.step()
        out["loss_%d" % it] = npy(res['loss'])
        for k, v in res['loss_dict'].items():
            out["ld%d_%s" % (it, k.replace('/', '_'))] = npy(v)
        out["gradnorm_%d" % it] = npy(gn)
        out["totalnorm_%d" % it] = npy(total_norm)
        out["psum_%d" % it] = npy(torch.stack([p.double().sum() for p in m.parameters()]))
        out["pabs_%d" % it] = npy(torch.stack([p.double().abs().sum() for p in m.parameters()]))
        # oracle in lock-step
        tot, ld, o_out, raw, norm = tr.step(data)
        gn_o = torch.stack([raw[k].norm() for k in tr.names])
        print("[%s] step %d: ref loss %.9f oracle %.9f | gradnorm rel dev %.2e | total norm %.6f vs %.6f" % (
            tag, it, float(res['loss']), float(tot), float(((gn - gn_o).abs() / (gn + 1e-12)).max()),
            float(total_norm), float(norm)))
        if it == 0:
            # re-run reference forward for tensors (BN state already advanced: use oracle's outputs vs hooks)
            pass
    # forward tensors at the initial state (fresh model, train mode)
    m2 = ref_model(H, W, with_pose)
    m2.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m2.train()
    data = O.synthetic_batch(B, H, W, seed=100)
    feats = m2.depth_backbone(data[('image', 0)])
    outs = m2.head.forward_depth(feats)
    for s in range(4):
        out["disp_%d" % s] = npy(outs[('disp', s)])
        out["depth_%d" % s] = npy(outs[('depth', s, s)])
    out["feat4"] = npy(feats[4])
    out["feat0_sub"] = npy(feats[0])[:, ::8, ::4, ::4]
    if with_pose:
        pf = m2.pose_backbone(torch.cat([data[('image', 0)], data[('image', 1)]], 1))
        aa, trn = m2.head.forward_pose([pf])
        out["axisangle_p"], out["translation_p"] = npy(aa), npy(trn)
    sdo = {k: v.clone() for k, v in sd0.items()}
    fo = O.resnet_forward(sdo, "depth_backbone.", data[('image', 0)])
    oo = O.depth_decoder_forward(sdo, "head.depth_decoder.", fo, 0.5, 100.0)
    print("[%s] oracle fwd dev: feat4 %.2e disp0 %.2e" % (tag, dev(fo[4], feats[4]), dev(oo[('disp', 0)], outs[('disp', 0)])))
    sd_ref = m.state_dict()
    print("[%s] after %d steps: max param dev oracle vs ref %.3e; running_mean dev %.3e" % (
        tag, steps, max(dev(tr.sd[k], sd_ref[k]) for k in tr.names),
        max(dev(tr.sd[k], sd_ref[k]) for k in sd_ref if k.endswith('running_mean'))))
    out["bn_rm_final"] = npy(torch.cat([sd_ref[k].flatten() for k in sd_ref if k.endswith('running_mean')]))
    out["bn_rv_final"] = npy(torch.cat([sd_ref[k].flatten() for k in sd_ref if k.endswith('running_var')]))
    np.savez_compressed(os.path.join(GOLD, "model_%s.npz" % tag), **out)


def gen_eval():
    """depth-error metrics of the REAL reference (monodepth_utils.compute_errors) on seeded inputs"""
    from oracle import eval_oracle as EO
    rng = np.random.RandomState(7)
    out = {}
    for k, n in enumerate((1, 7, 4096)):
        gt = (rng.rand(n).astype(np.float32) * 79 + 0.5)
        pred = (gt * np.exp(rng.randn(n).astype(np.float32) * 0.2)).astype(np.float32).clip(1e-3, 80.0)
        ref = np.array(mu.compute_errors(gt, pred), dtype=np.float64)
        mine = np.array(EO.compute_errors(gt, pred), dtype=np.float64)
        print("compute_errors case %d: oracle - reference max abs %.3e" % (k, np.abs(ref - mine).max()))
        out["gt_%d" % k], out["pred_%d" % k], out["err_%d" % k] = gt, pred, ref
    np.savez_compressed(os.path.join(GOLD, "eval.npz"), **out)


def gen_distill():
    """self-distillation stage through the REAL DistillWPoseMeta (monodepth2_model.py:150-206): loss terms and
    parameter-gradient norms on a seeded batch; cross-checks oracle/distill_oracle.py"""
    import tempfile
    from oracle import distill_oracle as D
    H, W, B = 64, 128, 2
    enc = dict(name='vision_base.networks.models.backbone.resnet.resnet', depth=18, pretrained=False, frozen_stages=-1,
               num_stages=4, out_indices=(-1, 0, 1, 2, 3), norm_eval=False, dilations=(1, 1, 1, 1))
    dec = dict(num_ch_enc=np.array([64, 64, 128, 256, 512]), num_output_channels=16, use_skips=True, scales=[0, 1, 2, 3],
               min_depth=0.5, max_depth=100)
    sd = D.init_states(seed=21, teacher_seed=22)
    with tempfile.TemporaryDirectory() as d:
        tpath = os.path.join(d, "teacher.pth")
        torch.save({k[len("teacher_net."):]: v.clone() for k, v in sd.items() if k.startswith("teacher_net.")}, tpath)
        m = build(name='monodepth.networks.models.meta_archs.monodepth2_model.DistillWPoseMeta',
                  teacher_net_cfg=EasyDict(name='monodepth.networks.models.meta_archs.teacher_model.MonoDepthInference',
                                           backbone_cfg=EasyDict(**enc),
                                           depth_head_cfg=EasyDict(name='monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder', **dec)),
                  teacher_net_path=tpath,
                  depth_backbone_cfg=EasyDict(**enc),
                  head_cfg=EasyDict(name='monodepth.networks.models.heads.monodepth2_decoder.MonoDepth2Decoder',
                                    scales=[0, 1, 2, 3], height=H, width=W, min_depth=0.5, max_depth=100.0,
                                    overlapped_mask=True, is_log_image=False, distillation_loss_weight=0.3,
                                    is_uncertain_distill=True,
                                    depth_decoder_cfg=EasyDict(name='monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoderUncertain', **dec)),
                  train_cfg=EasyDict(frame_ids=[0, 1, -1]), test_cfg=EasyDict())
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m.train()
    data = O.synthetic_batch(B, H, W, seed=400)
    torch.manual_seed(0)
    out = m(dict(data), dict(is_training=True, epoch_num=0, global_step=0))
    out["loss"].backward()
    names = [k for k, _ in m.named_parameters() if not k.startswith("teacher_net.")]
    gn = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for k, p in m.named_parameters()
                   if not k.startswith("teacher_net.")])
    res = {"B": B, "H": H, "W": W, "seed": 21, "teacher_seed": 22, "batch_seed": 400,
           "loss": float(out["loss"].detach()), "gradnorm": gn,
           "unc_w_grad": npy(dict(m.named_parameters())["head.depth_decoder.decoder.14.weight"].grad),
           "teacher_depth_0_mean": float(m.teacher_net.compute_teacher_depth(data[("image", 0)])[("teacher_depth", 0, 0)].mean())}
    for k, v in out["loss_dict"].items():
        res["ld_" + k.replace("/", "_")] = float(v)
    # oracle against the reference
    sdo = D.init_states(seed=21, teacher_seed=22)
    for k in sdo:
        if D.is_student_param(k):
            sdo[k].requires_grad_(True)
    tot, losses, _ = D.forward_train(sdo, O.synthetic_batch(B, H, W, seed=400))
    tot.backward()
    gno = np.array([float(sdo[k].grad.norm()) for k in names])
    print("distill: loss ref %.8f oracle %.8f | gradnorm max rel dev %.3e" % (
        res["loss"], float(tot), np.abs(gno - gn).max() / gn.max()))
    for k in losses:
        if k.startswith("distilation"):
            print("   %s ref %.6f oracle %.6f" % (k, res["ld_" + k.replace("/", "_")], float(losses[k])))
    np.savez_compressed(os.path.join(GOLD, "distill.npz"), **res)


def gen_augment():
    """Training input pipeline (configs/kitti_wpose_example:129-155) through the REAL reference classes, built by
    the reference's own builder.  cv2 is absent here: tools/ref_shims/cv2 routes warpAffine / cvtColor to the oracle's
    restatement, so this pins draw order, P2 / pose bookkeeping, mirror, colour arithmetic and Normalize — not OpenCV."""
    from oracle import augment_oracle as A
    aug = 'vision_base.data.augmentations.augmentations'
    frame_idxs = [0, 1, -1]
    out_h, out_w, H, W = 24, 80, 300, 420
    resize_keys = [('image', i) for i in frame_idxs] + [('original_image', i) for i in frame_idxs]
    colour_keys = [('image', i) for i in frame_idxs]
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    seeds = dict(warp=101, bright=102, contrast=103, sat=104)
    E = EasyDict
    cfg = E(name='vision_base.utils.builder.Sequential', cfg_list=[
        E(name=aug + '.ConvertToFloat'),
        E(name=aug + '.RandomWarpAffine', output_w=out_w, output_h=out_h, random_seed=seeds['warp']),
        E(name=aug + '.RandomMirror', mirror_prob=0.5, pose_axis_pairs=[(("relative_pose", i), 0) for i in frame_idxs[1:]]),
        E(name='vision_base.utils.builder.Shuffle', cfg_list=[
            E(name=aug + '.RandomBrightness', distort_prob=1.0, random_seed=seeds['bright']),
            E(name=aug + '.RandomContrast', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=seeds['contrast']),
            E(name='vision_base.utils.builder.Sequential', cfg_list=[
                E(name=aug + '.ConvertColor', transform='HSV'),
                E(name=aug + '.RandomSaturation', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=seeds['sat']),
                E(name=aug + '.ConvertColor', current='HSV', transform='RGB')])],
          image_keys=colour_keys),
        E(name=aug + '.Normalize', mean=mean, stds=std, image_keys=colour_keys),
        E(name=aug + '.Normalize', mean=np.array([0, 0, 0]), stds=np.array([1, 1, 1]),
          image_keys=[('original_image', i) for i in frame_idxs]),
        E(name=aug + '.ConvertToTensor')],
        image_keys=resize_keys, calib_keys=['P2'], gt_image_keys=['patched_mask'])
    transform = build(**cfg)
    # the oracle draws from generators seeded like the reference instances
    rngs = {k: np.random.default_rng(v) for k, v in seeds.items()}
    res = dict(H=H, W=W, out_h=out_h, out_w=out_w, mean=mean, std=std, n=4, frame_seed=900, global_seed=77,
               **{"seed_" + k: v for k, v in seeds.items()})
    np.random.seed(77)
    gstate = np.random.get_state()
    maxdev = 0.0
    for n in range(4):
        fr = np.random.RandomState(900 + n)
        frames = [fr.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in frame_idxs]
        P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)
        poses = []
        for j in range(2):
            from scipy.spatial.transform import Rotation as R
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = R.from_euler('xyz', fr.uniform(-0.05, 0.05, 3)).as_matrix()
            T[:3, 3] = fr.uniform(-1, 1, 3)
            poses.append(T)
        data = {}
        for i, f in zip(frame_idxs, frames):
            data[('image', i)] = f.copy()
            data[('original_image', i)] = f.copy()
        data['patched_mask'] = np.ones([H, W])
        data['P2'] = P2.copy()
        for i, T in zip(frame_idxs[1:], poses):
            data[('relative_pose', i)] = T.copy()
        np.random.set_state(gstate)
        out = transform(data)
        # the same sample through the oracle, with the global stream rewound to the same point
        np.random.set_state(gstate)
        plan = A.draw_plan(rngs['warp'], rngs['bright'], rngs['contrast'], rngs['sat'], H, W, out_w, out_h)
        gstate = np.random.get_state()
        imgs, origs, mask = A.run_sample(frames, plan, out_w, out_h, mean, std)
        P = A.warp_P2(P2, plan["final_scale"], plan["shift_w"], plan["shift_h"])
        if plan["mirror"]:
            P = A.mirror_P2(P, out_w)
        for j, i in enumerate(frame_idxs):
            res["s%d_image_%d" % (n, j)] = npy(out[('image', i)])
            res["s%d_orig_%d" % (n, j)] = npy(out[('original_image', i)])
            maxdev = max(maxdev, float(np.abs(npy(out[('image', i)]) - imgs[j]).max()),
                         float(np.abs(npy(out[('original_image', i)]) - origs[j]).max()))
        res["s%d_mask" % n] = npy(out['patched_mask'])
        res["s%d_P2" % n] = np.asarray(out['P2'])
        res["s%d_mirror" % n] = plan["mirror"]
        res["s%d_order" % n] = plan["order"]
        for j, i in enumerate(frame_idxs[1:]):
            res["s%d_pose_%d" % (n, j)] = np.asarray(out[('relative_pose', i)])
            want = A.flip_relative_pose(poses[j].copy(), 0) if plan["mirror"] else poses[j]
            maxdev = max(maxdev, float(np.abs(want - res["s%d_pose_%d" % (n, j)]).max()))
            res["s%d_pose_in_%d" % (n, j)] = poses[j]
        maxdev = max(maxdev, float(np.abs(mask - res["s%d_mask" % n]).max()), float(np.abs(P - res["s%d_P2" % n]).max()))
        print("augment sample %d: mirror=%s order=%s" % (n, plan["mirror"], plan["order"]))
    print("augment: oracle vs reference classes, max abs deviation over images/masks/P2/poses: %.3e" % maxdev)
    # validation input: ConvertToFloat, Resize, Normalize, ConvertToTensor (configs/kitti_wpose_example:156-166)
    if not hasattr(np, "int"):
        np.int = int          # the reference's Resize uses the alias numpy removed in 1.24 (:134, :157)
    vdev = 0.0
    for tag, kw, shp in (("stretch", dict(preserve_aspect_ratio=False), (75, 250)),
                         ("pad1", dict(preserve_aspect_ratio=True, force_pad=True), (120, 250)),
                         ("pad0", dict(preserve_aspect_ratio=True, force_pad=True), (60, 400)),
                         ("crop1", dict(preserve_aspect_ratio=True, force_pad=False), (60, 400))):
        vt = build(**E(name='vision_base.utils.builder.Sequential', cfg_list=[
            E(name=aug + '.ConvertToFloat'), E(name=aug + '.Resize', size=(48, 160), **kw),
            E(name=aug + '.Normalize', mean=mean, stds=std), E(name=aug + '.ConvertToTensor')],
            image_keys=[('image', 0)], calib_keys=['P2']))
        fr = np.random.RandomState(950 + len(tag))
        frame = fr.randint(0, 256, size=shp + (3,)).astype(np.uint8)
        P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)
        out = vt({('image', 0): frame.copy(), 'P2': P2.copy()})
        res["val_%s_image" % tag] = npy(out[('image', 0)])
        res["val_%s_P2" % tag] = np.asarray(out['P2'])
        res["val_%s_shape" % tag] = np.array(shp)
        res["val_%s_seed" % tag] = 950 + len(tag)
        res["val_%s_effective" % tag] = np.asarray(out[('image_resize', 'effective_size')])
        res["val_%s_original" % tag] = np.asarray(out[('image_resize', 'original_shape')])
        vdev = max(vdev, float(np.abs(A.run_val_sample(frame, (48, 160), mean, std, **kw) - res["val_%s_image" % tag]).max()))
    print("augment (validation Resize): oracle vs reference class, max abs deviation %.3e" % vdev)
    np.savez_compressed(os.path.join(GOLD, "augment.npz"), **res)


def gen_augment_resize():
    """Training input of the Resize-based shipped configs (configs/multi_dataset_example:178-205) through the REAL
    reference classes: Resize of six frames + nearest patched_mask, Shuffle of the colour ops, RandomMirror AFTER
    them, two Normalizes.  cv2.resize is the shim's restatement (unpinned, like warpAffine)."""
    from oracle import augment_oracle as A
    if not hasattr(np, "int"):
        np.int = int
    aug = 'vision_base.data.augmentations.augmentations'
    frame_idxs = [0, 1, -1]
    size = (48, 160)
    resize_keys = [('image', i) for i in frame_idxs] + [('original_image', i) for i in frame_idxs]
    colour_keys = [('image', i) for i in frame_idxs]
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    seeds = dict(bright=202, contrast=203, sat=204)
    E = EasyDict
    cfg = E(name='vision_base.utils.builder.Sequential', cfg_list=[
        E(name=aug + '.ConvertToFloat'),
        E(name=aug + '.Resize', size=size, preserve_aspect_ratio=True, force_pad=True),
        E(name='vision_base.utils.builder.Shuffle', cfg_list=[
            E(name=aug + '.RandomBrightness', distort_prob=1.0, random_seed=seeds['bright']),
            E(name=aug + '.RandomContrast', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=seeds['contrast']),
            E(name='vision_base.utils.builder.Sequential', cfg_list=[
                E(name=aug + '.ConvertColor', transform='HSV'),
                E(name=aug + '.RandomSaturation', distort_prob=1.0, lower=0.6, upper=1.4, random_seed=seeds['sat']),
                E(name=aug + '.ConvertColor', current='HSV', transform='RGB')])],
          image_keys=colour_keys),
        E(name=aug + '.RandomMirror', mirror_prob=0.5, pose_axis_pairs=[(("relative_pose", i), 0) for i in frame_idxs[1:]]),
        E(name=aug + '.Normalize', mean=mean, stds=std, image_keys=colour_keys),
        E(name=aug + '.Normalize', mean=np.array([0, 0, 0]), stds=np.array([1, 1, 1]),
          image_keys=[('original_image', i) for i in frame_idxs]),
        E(name=aug + '.ConvertToTensor')],
        image_keys=resize_keys, calib_keys=['P2'], gt_image_keys=['patched_mask'])
    transform = build(**cfg)
    rngs = {k: np.random.default_rng(v) for k, v in seeds.items()}
    shapes = [(75, 250), (120, 250), (60, 400), (96, 320)]      # pad_1, pad_1, pad_0, exact fit
    res = dict(size=np.array(size), mean=mean, std=std, n=len(shapes), frame_seed=1200, global_seed=88,
               shapes=np.array(shapes), **{"seed_" + k: v for k, v in seeds.items()})
    np.random.seed(88)
    gstate = np.random.get_state()
    maxdev = 0.0
    for n, (H, W) in enumerate(shapes):
        fr = np.random.RandomState(1200 + n)
        frames = [fr.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in frame_idxs]
        P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]], dtype=np.float64)
        from scipy.spatial.transform import Rotation as R
        poses = []
        for j in range(2):
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = R.from_euler('xyz', fr.uniform(-0.05, 0.05, 3)).as_matrix()
            T[:3, 3] = fr.uniform(-1, 1, 3)
            poses.append(T)
        data = {}
        for i, f in zip(frame_idxs, frames):
            data[('image', i)] = f.copy()
            data[('original_image', i)] = f.copy()
        data['patched_mask'] = np.ones([H, W])
        data['P2'] = P2.copy()
        for i, T in zip(frame_idxs[1:], poses):
            data[('relative_pose', i)] = T.copy()
        np.random.set_state(gstate)
        out = transform(data)
        np.random.set_state(gstate)
        plan = A.draw_resize_plan(rngs['bright'], rngs['contrast'], rngs['sat'])
        gstate = np.random.get_state()
        imgs, origs, mask, syx = A.run_resize_train_sample(frames, plan, size, mean, std)
        P = P2.copy()
        P[0, :] *= syx[1]; P[1, :] *= syx[0]
        if plan["mirror"]:
            P = A.mirror_P2(P, size[1])
        for j, i in enumerate(frame_idxs):
            res["s%d_image_%d" % (n, j)] = npy(out[('image', i)])
            res["s%d_orig_%d" % (n, j)] = npy(out[('original_image', i)])
            maxdev = max(maxdev, float(np.abs(npy(out[('image', i)]) - imgs[j]).max()),
                         float(np.abs(npy(out[('original_image', i)]) - origs[j]).max()))
        res["s%d_mask" % n] = npy(out['patched_mask'])
        res["s%d_P2" % n] = np.asarray(out['P2'])
        res["s%d_mirror" % n] = plan["mirror"]
        res["s%d_order" % n] = plan["order"]
        for j, i in enumerate(frame_idxs[1:]):
            res["s%d_pose_%d" % (n, j)] = np.asarray(out[('relative_pose', i)])
            res["s%d_pose_in_%d" % (n, j)] = poses[j]
        maxdev = max(maxdev, float(np.abs(mask - res["s%d_mask" % n]).max()), float(np.abs(P - res["s%d_P2" % n]).max()))
        print("augment-resize sample %d: shape %s mirror=%s order=%s" % (n, (H, W), plan["mirror"], plan["order"]))
    print("augment-resize: oracle vs reference classes, max abs deviation %.3e" % maxdev)
    np.savez_compressed(os.path.join(GOLD, "augment_resize.npz"), **res)


def gen_fisheye():
    """Fisheye path (BASELINE configs[3]) through the REAL classes: MeiCameraProjection LUT + cam2image
    (mei_fisheye_utils.py:14-187) and FishEyeDecoder.loss with gradients (monodepth2_decoder.py:350-420)."""
    from monodepth.networks.utils.mei_fisheye_utils import MeiCameraProjection
    from oracle import fisheye_oracle as FO
    out = {}
    proj = MeiCameraProjection()
    # --- LUT at 48x48 for two calibrations ---
    H = W = 48
    for v in range(2):
        P, calib = FO.synthetic_calib(H, W, v)
        norm = torch.ones(1, 1, H, W)
        pts, mask = proj.image2cam(norm, P[None], [calib])
        lut = np.stack([npy(pts[0, 0, ..., 0]), npy(pts[0, 0, ..., 1]), npy(pts[0, 0, ..., 2]), npy(mask[0, 0])], 0)
        out["lut%d" % v] = lut
        out["lut%d_P" % v] = npy(P)
        out["lut%d_calib" % v] = np.array([calib["distortion_parameters"]["k1"], calib["distortion_parameters"]["k2"],
                                           calib["mirror_parameters"]["xi"]])
        mine = np.stack(FO.mei_lut(H, W, P[0, 0].item(), P[1, 1].item(), P[0, 2].item(), P[1, 2].item(),
                                   calib["distortion_parameters"]["k1"], calib["distortion_parameters"]["k2"],
                                   calib["mirror_parameters"]["xi"]), 0)
        print("fisheye LUT %d: valid %.1f %%, oracle max dev %.3e, mask mismatches %d" % (
            v, 100 * lut[3].mean(), np.abs(mine[:3] - lut[:3]).max(), int((mine[3] != lut[3]).sum())))
    # --- cam2image on random points in front of / beside the camera ---
    g = torch.Generator().manual_seed(5)
    pts = torch.randn(64, 3, generator=g) * torch.tensor([3.0, 3.0, 2.0]) + torch.tensor([0.0, 0.0, 2.5])
    P, calib = FO.synthetic_calib(H, W, 0)
    uvz = proj.cam2image(pts, P, calib)
    out["c2i_points"], out["c2i_uvz"] = npy(pts), npy(uvz)
    u, v_ = FO.cam2image(pts, P, calib)
    print("fisheye cam2image oracle dev", dev(torch.stack([u, v_], -1), uvz[..., :2]))
    # --- loss chain with gradients ---
    B, H, W = 2, 64, 64
    data = O.synthetic_batch(B, H, W, seed=31)
    data['patched_mask'][:, :5, :] = 0
    Ps, calibs = zip(*[FO.synthetic_calib(H, W, v) for v in range(B)])
    data["P2"] = torch.stack(Ps, 0)
    data["calib_meta"] = list(calibs)
    g = torch.Generator().manual_seed(4)
    dec = build(name='monodepth.networks.models.heads.monodepth2_decoder.FishEyeDecoder', scales=[0, 1, 2, 3],
                height=H, width=W, frame_ids=[0, 1, -1], min_depth=0.5, max_depth=150.0, overlapped_mask=True,
                is_log_image=False,
                depth_decoder_cfg=dict(name='monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder',
                                       num_ch_enc=np.array([64, 64, 128, 256, 512]), num_output_channels=64,
                                       use_skips=True, scales=[0, 1, 2, 3], min_depth=0.5, max_depth=150))
    outputs, leaves = {}, {}
    for s in range(4):
        h, w = H >> s, W >> s
        ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
        d = (5 + 20 * (1 - ys) + 3 * torch.rand(B, 1, h, w, generator=g)).requires_grad_(True)
        leaves[("depth", s)] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = mu.depth_to_disp(d, 0.5, 150.0)
    for f in (1, -1):
        aa = (0.01 * torch.randn(B, 1, 3, generator=g)).requires_grad_(True)
        tr = torch.tensor([[[0.6 if f > 0 else -0.6, -0.01, 0.03]]]).repeat(B, 1, 1) + 0.02 * torch.randn(B, 1, 3, generator=g)
        tr.requires_grad_(True)
        leaves[("aa", f)], leaves[("tr", f)] = aa, tr
        outputs[("cam_T_cam", f)] = mu.transformation_from_parameters(aa, tr, invert=(f < 0))
    torch.manual_seed(0)
    res = dec.loss(outputs, data)
    res['loss'].backward()
    out.update(H=H, W=W, seed=31, total_loss=npy(res['loss']))
    for k, v in res['loss_dict'].items():
        out["ld_" + k.replace('/', '_')] = npy(v)
    for s in range(4):
        out["depth_%d" % s] = npy(leaves[("depth", s)])
        out["gdepth_%d" % s] = npy(leaves[("depth", s)].grad)
    for f in (1, -1):
        tag = "p" if f > 0 else "m"
        out["aa_" + tag], out["tr_" + tag] = npy(leaves[("aa", f)]), npy(leaves[("tr", f)])
        out["gaa_" + tag], out["gtr_" + tag] = npy(leaves[("aa", f)].grad), npy(leaves[("tr", f)].grad)
        out["warp0_" + tag] = npy(outputs[("original_image", f, 0)])[:, :, ::2, ::2]
        out["ovmask0_" + tag] = npy(outputs[("overlapped_mask", f, 0)])
    for f in (0, 1, -1):
        out["img_%s" % {0: "0", 1: "p", -1: "m"}[f]] = npy(data[("original_image", f)])
    out["P2"] = npy(data["P2"]); out["patched_mask"] = npy(data["patched_mask"])
    out["calib"] = np.array([[c["distortion_parameters"]["k1"], c["distortion_parameters"]["k2"],
                              c["mirror_parameters"]["xi"]] for c in calibs])
    pred = dec.get_prediction(data, {("depth", 0, 0): leaves[("depth", 0)].detach()})
    out["pred_depth"] = npy(pred["depth"])
    # cross-check the oracle
    o2, lv = {}, {}
    for s in range(4):
        d = leaves[("depth", s)].detach().clone().requires_grad_(True); lv[s] = d
        o2[("depth", s, s)] = d; o2[("disp", s)] = O.depth_to_disp(d, 0.5, 150.0)
    for f in (1, -1):
        o2[("cam_T_cam", f)] = outputs[("cam_T_cam", f)].detach()
    tot, ld = FO.photometric_loss(o2, data)
    tot.backward()
    print("fisheye chain: ref loss %.9f oracle %.9f | gdepth0 dev rel %.3e | warp dev %.3e | ov mismatches %d" % (
        float(res['loss']), float(tot),
        dev(lv[0].grad, leaves[("depth", 0)].grad) / float(leaves[("depth", 0)].grad.abs().max()),
        dev(o2[("original_image", 1, 0)], outputs[("original_image", 1, 0)]),
        int((o2[("overlapped_mask", 1, 0)] != outputs[("overlapped_mask", 1, 0)]).sum())))
    np.savez_compressed(os.path.join(GOLD, "fisheye.npz"), **out)


def gen_model_r50fx():
    """BASELINE configs[4] wiring at a small size: MonoDepthWPose, ResNet-50 (Bottleneck), 64 depth bins,
    base_fx = 492 (configs/multi_dataset_example:227-257) with per-sample focal lengths; forward tensors, loss_dict and
    per-parameter gradient norms of one step from the REAL reference."""
    B, H, W = 2, 64, 128
    enc = [64, 256, 512, 1024, 2048]
    bb = dict(name='vision_base.networks.models.backbone.resnet.resnet', depth=50, pretrained=False,
              frozen_stages=-1, num_stages=4, out_indices=(-1, 0, 1, 2, 3), norm_eval=False, dilations=(1, 1, 1, 1))
    head = dict(name='monodepth.networks.models.heads.monodepth2_decoder.MonoDepth2Decoder', scales=[0, 1, 2, 3],
                height=H, width=W, min_depth=0.5, max_depth=100.0, overlapped_mask=True, is_log_image=False,
                depth_decoder_cfg=dict(name='monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder',
                                       num_ch_enc=np.array(enc), num_output_channels=64, use_skips=True,
                                       scales=[0, 1, 2, 3], min_depth=0.5, max_depth=100, base_fx=492))
    m = build(name='monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthWPose', depth_backbone_cfg=bb,
              head_cfg=head, train_cfg=EasyDict(frame_ids=[0, 1, -1]), test_cfg=EasyDict())
    sd0 = O.init_state(seed=7, depth=50, with_pose=False, num_out=64)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m.train()
    data = O.synthetic_batch(B, H, W, seed=300)
    data["P2"][0, 0, 0] *= 1.4          # two focal lengths in the batch: depth scales 0.211 and 0.151
    data["P2"][1, 0, 0] *= 1.0
    torch.manual_seed(0)
    res = m(dict(data), dict(is_training=True))
    res['loss'].mean().backward()
    out = {"B": B, "H": H, "W": W, "init_seed": 7, "batch_seed": 300, "base_fx": 492.0, "fx_mul": np.array([1.4, 1.0]),
           "loss": npy(res['loss'])}
    for k, v in res['loss_dict'].items():
        out["ld_" + k.replace('/', '_')] = npy(v)
    out["gradnorm"] = npy(torch.stack([p.grad.norm() for p in m.parameters()]))
    m2 = build(name='monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthWPose', depth_backbone_cfg=bb,
               head_cfg=head, train_cfg=EasyDict(frame_ids=[0, 1, -1]), test_cfg=EasyDict())
    m2.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m2.train()
    feats = m2.depth_backbone(data[('image', 0)])
    outs = m2.head.forward_depth(feats, data['P2'])
    for s in range(4):
        out["disp_%d" % s] = npy(outs[('disp', s)])
        out["depth_%d" % s] = npy(outs[('depth', s, s)])
    out["feat4"] = npy(feats[4])[:, ::16]
    tr = O.OracleTrainer(sd0, depth=50, with_pose=False, base_fx=492.0)
    tot, ld, o_out, raw, norm = tr.step(data)
    gn_o = torch.stack([raw[k].norm() for k in tr.names])
    gn = torch.from_numpy(out["gradnorm"])
    print("[r50fx] ref loss %.9f oracle %.9f | gradnorm rel dev %.2e | depth0 dev %.2e" % (
        float(res['loss']), float(tot), float(((gn - gn_o).abs() / (gn + 1e-12)).max()),
        dev(o_out[('depth', 0, 0)], outs[('depth', 0, 0)])))
    np.savez_compressed(os.path.join(GOLD, "model_r50fx.npz"), **out)


def gen_kitti_dataset():
    """the REAL KittiDepthMonoDataset (mono_dataset.py:108-250) over the seeded fake KITTI tree of
    tests/helpers_kitti.py: filtered index, relative poses, P2, image tensors"""
    import tempfile
    from tests import helpers_kitti as HK
    from monodepth.data.datasets.mono_dataset import KittiDepthMonoDataset
    out = {}
    with tempfile.TemporaryDirectory() as d:
        raw, split = HK.make_tree(d, seed=5)
        ds = KittiDepthMonoDataset(**HK.dataset_cfg(raw, split, prefix=''))
        out["n"] = len(ds)
        out["index"] = np.array([[o["index"], 0 if o["side"] == "l" else 1] for o in ds.imdb])
        for i in range(len(ds)):
            smp = ds[i]
            for f, tag in ((0, "0"), (1, "p"), (-1, "m")):
                out["s%d_image_%s" % (i, tag)] = npy(smp[("image", f)])
                out["s%d_orig_%s" % (i, tag)] = npy(smp[("original_image", f)])
            out["s%d_pose_p" % i], out["s%d_pose_m" % i] = np.asarray(smp[("relative_pose", 1)]), np.asarray(smp[("relative_pose", -1)])
            out["s%d_P2" % i], out["s%d_original_P2" % i] = np.asarray(smp["P2"]), np.asarray(smp["original_P2"])
            out["s%d_mask" % i] = npy(smp["patched_mask"])
    print("kitti dataset: %d of 5 split entries kept, sides %s" % (out["n"], out["index"][:, 1].tolist()))
    np.savez_compressed(os.path.join(GOLD, "kitti_dataset.npz"), **out)


def gen_loss_options():
    """optional terms of MonoDepth2Decoder.loss no shipped config enables: precomputed motion_mask
    (monodepth2_decoder.py:243-246) and the pose L1 term (:176-183, 322-326), from the REAL decoder"""
    B, H, W = 2, 64, 96
    data = O.synthetic_batch(B, H, W, seed=41)
    g = torch.Generator().manual_seed(8)
    mm = (torch.rand(B, H // 8, W // 8, generator=g) > 0.6).float()
    data["motion_mask"] = torch.nn.functional.interpolate(mm[:, None], size=(H, W), mode="nearest")[:, 0]   # blocky 0/1
    dec = ref_model(H, W, True).head
    dec.pose_loss_weight = 0.5
    outputs, leaves = {}, {}
    for s in range(4):
        h, w = H >> s, W >> s
        ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
        d = (4 + 25 * (1 - ys) + 3 * torch.rand(B, 1, h, w, generator=g)).requires_grad_(True)
        leaves[("depth", s)] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = mu.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        aa = (0.01 * torch.randn(B, 1, 3, generator=g)).requires_grad_(True)
        tr = (torch.tensor([[[0.02, -0.01, -0.6 if f > 0 else 0.6]]]).repeat(B, 1, 1) + 0.02 * torch.randn(B, 1, 3, generator=g)).requires_grad_(True)
        leaves[("aa", f)], leaves[("tr", f)] = aa, tr
        outputs[("cam_T_cam", f)] = mu.transformation_from_parameters(aa, tr, invert=(f < 0))
    res = dec.loss(outputs, data)
    res['loss'].backward()
    out = {"H": H, "W": W, "seed": 41, "pose_loss_weight": 0.5, "total_loss": npy(res['loss']), "motion_mask": npy(data["motion_mask"])}
    for k, v in res['loss_dict'].items():
        out["ld_" + k.replace('/', '_')] = npy(v)
    for s in range(4):
        out["depth_%d" % s] = npy(leaves[("depth", s)])
        out["gdepth_%d" % s] = npy(leaves[("depth", s)].grad)
    for f in (1, -1):
        tag = "p" if f > 0 else "m"
        out["aa_" + tag], out["tr_" + tag] = npy(leaves[("aa", f)]), npy(leaves[("tr", f)])
        out["gaa_" + tag], out["gtr_" + tag] = npy(leaves[("aa", f)].grad), npy(leaves[("tr", f)].grad)
        out["pose_" + tag] = npy(data[("relative_pose", f)])
    # oracle cross-check
    o2, lv = {}, {}
    for s in range(4):
        d = leaves[("depth", s)].detach().clone().requires_grad_(True); lv[s] = d
        o2[("depth", s, s)] = d; o2[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        o2[("cam_T_cam", f)] = outputs[("cam_T_cam", f)].detach()
    tot, ld = O.photometric_loss(o2, data)
    pl = sum((data[("relative_pose", f)] - o2[("cam_T_cam", f)]).abs().mean() for f in (1, -1))
    (tot + 0.5 * pl).backward()
    print("loss options: ref total %.9f oracle %.9f | gdepth0 rel dev %.2e" % (
        float(res['loss']), float(tot + 0.5 * pl),
        dev(lv[0].grad, leaves[("depth", 0)].grad) / float(leaves[("depth", 0)].grad.abs().max())))
    np.savez_compressed(os.path.join(GOLD, "loss_options.npz"), **out)


def no_overlap_case(H=64, W=96, B=2, seed=43):
    """inputs of the overlapped_mask=False golden: large translations, so that a good part of the samples leaves the
    source frames (those are exactly the pixels the option changes)"""
    data = O.synthetic_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(18)
    depths = []
    for s in range(4):
        h, w = H >> s, W >> s
        ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
        depths.append(4 + 25 * (1 - ys) + 3 * torch.rand(B, 1, h, w, generator=g))
    poses = {}
    for f in (1, -1):
        aa = 0.03 * torch.randn(B, 1, 3, generator=g)
        tr = torch.tensor([[[0.9 if f > 0 else -0.9, -0.05, -0.8 if f > 0 else 0.8]]]).repeat(B, 1, 1) + 0.02 * torch.randn(B, 1, 3, generator=g)
        poses[f] = (aa, tr)
    return data, depths, poses


def gen_no_overlap_mask():
    """MonoDepth2Decoder.loss with overlapped_mask=False (configs/multi_dataset_example, nusc_wpose_example;
    monodepth2_decoder.py:110-116, 230-235) from the REAL decoder: totals, depth and pose gradients"""
    H, W = 64, 96
    data, depths, poses = no_overlap_case(H, W)
    dec = ref_model(H, W, True).head
    dec.overlapped_mask = False
    outputs, leaves = {}, {}
    for s in range(4):
        d = depths[s].clone().requires_grad_(True)
        leaves[("depth", s)] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = mu.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        aa, tr = poses[f][0].clone().requires_grad_(True), poses[f][1].clone().requires_grad_(True)
        leaves[("aa", f)], leaves[("tr", f)] = aa, tr
        outputs[("cam_T_cam", f)] = mu.transformation_from_parameters(aa, tr, invert=(f < 0))
    torch.manual_seed(0)
    res = dec.loss(outputs, data)
    res['loss'].backward()
    assert not any(k[0] == "overlapped_mask" for k in outputs)
    out = {"H": H, "W": W, "total_loss": npy(res['loss'])}
    for k, v in res['loss_dict'].items():
        out["ld_" + k.replace('/', '_')] = npy(v)
    for s in range(4):
        out["gdepth_%d" % s] = npy(leaves[("depth", s)].grad)
    for f in (1, -1):
        tag = "p" if f > 0 else "m"
        out["gaa_" + tag], out["gtr_" + tag] = npy(leaves[("aa", f)].grad), npy(leaves[("tr", f)].grad)
    # how much the option matters on this case: the same inputs with the mask on
    dec.overlapped_mask = True
    o_on = {k: v.detach() for k, v in outputs.items() if k[0] in ("depth", "disp", "cam_T_cam")}
    torch.manual_seed(0)
    on = dec.loss(o_on, data)
    out["total_loss_masked"] = npy(on['loss'])
    out["outside_frac"] = np.float64(1.0 - float(torch.stack([o_on[("overlapped_mask", f, 0)].float().mean() for f in (1, -1)]).mean()))
    o2, lv = {}, {}
    for s in range(4):
        d = depths[s].clone().requires_grad_(True); lv[s] = d
        o2[("depth", s, s)] = d; o2[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f in (1, -1):
        o2[("cam_T_cam", f)] = outputs[("cam_T_cam", f)].detach()
    tot, ld = O.photometric_loss(o2, data, overlapped_mask=False)
    tot.backward()
    print("no overlap mask: ref total %.9f (masked %.9f, %.1f %% of the samples outside) oracle %.9f | gdepth0 rel dev %.2e" % (
        float(res['loss']), float(on['loss']), 100 * float(out["outside_frac"]), float(tot),
        dev(lv[0].grad, leaves[("depth", 0)].grad) / float(leaves[("depth", 0)].grad.abs().max())))
    np.savez_compressed(os.path.join(GOLD, "no_overlap_mask.npz"), **out)


def thin(t, limit=4096):
    """every k-th element of the flattened tensor (k = the smallest stride that leaves <= `limit` values), float32;
    tests apply the same rule to what they compare"""
    a = t.detach().reshape(-1)
    k = max(1, -(-a.numel() // limit))
    return a[::k].to(torch.float32).numpy().copy()


def sigmoid_decoder_case():
    """seeded inputs of the sigmoid-disparity DepthDecoder golden (shared with tests/)"""
    g = torch.Generator().manual_seed(77)
    B, H, W = 2, 64, 128
    chans = [64, 64, 128, 256, 512]
    feats = [torch.randn(B, c, H >> (k + 1), W >> (k + 1), generator=g) * 0.5 for k, c in enumerate(chans)]
    sd = {k[len("head.depth_decoder."):]: v for k, v in O.init_state(seed=9, with_pose=False, num_out=1).items()
          if k.startswith("head.depth_decoder.")}
    P2 = torch.tensor([[[700.0, 0, 64, 0], [0, 700, 32, 0], [0, 0, 1, 0]], [[540.0, 0, 60, 0], [0, 540, 30, 0], [0, 0, 1, 0]]])
    wd = [torch.randn(B, 1, H >> s, W >> s, generator=g) for s in range(4)]
    wq = [torch.randn(B, 1, H >> s, W >> s, generator=g) for s in range(4)]
    return feats, sd, P2, wd, wq


def gen_sigmoid_decoder():
    """The base-class DepthDecoder (sigmoid disparity, depth_encoder.py:17-111) with and without base_fx: outputs,
    feature gradients and parameter-gradient norms of sum_s <depth_s, wd_s> * 1e-2 + <disp_s, wq_s>."""
    from monodepth.networks.models.heads.depth_encoder import DepthDecoder
    feats, sd, P2, wd, wq = sigmoid_decoder_case()
    out = {}
    for tag, base_fx in (("plain", None), ("fx", 600.0)):
        m = DepthDecoder(num_ch_enc=np.array([64, 64, 128, 256, 512]), scales=[0, 1, 2, 3], num_output_channels=1,
                         use_skips=True, min_depth=0.5, max_depth=100, base_fx=base_fx)
        m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        m.train()
        fl = [f.clone().requires_grad_(True) for f in feats]
        res = m(fl, P2 if base_fx is not None else None)
        loss = sum((res[("depth", s, s)] * wd[s]).sum() * 1e-2 + (res[("disp", s)] * wq[s]).sum() for s in range(4))
        loss.backward()
        o = O.depth_decoder_forward({("d." + k): v for k, v in sd.items()}, "d.", [f.clone() for f in feats], 0.5, 100.0,
                                    P2=P2 if base_fx is not None else None, base_fx=base_fx, sigmoid=True)
        print("sigmoid decoder [%s]: oracle depth dev %.2e disp dev %.2e" % (
            tag, dev(o[("depth", 0, 0)], res[("depth", 0, 0)]), dev(o[("disp", 0)], res[("disp", 0)])))
        out[tag + "_loss"] = npy(loss)
        for s in range(4):
            out["%s_depth%d" % (tag, s)] = thin(res[("depth", s, s)])
            out["%s_disp%d" % (tag, s)] = thin(res[("disp", s)])
        for k, f in enumerate(fl):
            out["%s_gfeat%d" % (tag, k)] = thin(f.grad)
        out[tag + "_gnorm"] = npy(torch.stack([p.grad.norm() for p in m.parameters()]))
    np.savez_compressed(os.path.join(GOLD, "sigmoid_decoder.npz"), **out)


def gen_frozen():
    """ResNet.train() with frozen_stages / norm_eval (resnet.py:169-197) inside the training step of the reference's
    own hook: 3 Adam steps of the dataset-pose meta-arch, with (a) stem + layer1 frozen, (b) every BatchNorm in eval
    mode.  Losses, gradient norms, parameter sums per step, final running statistics."""
    from vision_base.pipeline_hooks.train_val_hooks.base_training_hooks import BaseTrainingHook
    B, H, W = 2, 64, 128
    out = {}
    for tag, fs, ne in (("fs1", 1, False), ("ne", -1, True)):
        sd0 = O.init_state(seed=3, with_pose=False)
        # running statistics away from (0, 1): an eval-mode BatchNorm must really use them
        g = torch.Generator().manual_seed(5)
        for k in sd0:
            if k.endswith("running_mean"):
                sd0[k] = 0.1 * torch.randn(sd0[k].shape, generator=g)
            elif k.endswith("running_var"):
                sd0[k] = 0.5 + torch.rand(sd0[k].shape, generator=g)
        m = ref_model(H, W, False, frozen_stages=fs, norm_eval=ne)
        m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        hook = BaseTrainingHook(clip_gradients=35.0)
        tr = O.OracleTrainer(sd0, with_pose=False, frozen_stages=fs, norm_eval=ne)
        for it in range(3):
            data = O.synthetic_batch(B, H, W, seed=300 + it)
            # the hook's body (base_training_hooks.py:30-49) without its .cuda() calls
            opt.zero_grad()
            torch.manual_seed(0)
            res = m(dict(data), dict(is_training=True))
            res['loss'].mean().backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), hook.clip_gradients)
            opt.step()
            out["%s_loss_%d" % (tag, it)] = npy(res['loss'])
            out["%s_psum_%d" % (tag, it)] = npy(torch.stack([p.double().sum() for p in m.parameters()]))
            tot, ld, o_out, raw, norm = tr.step(data)
            print("[frozen %s] step %d: ref loss %.9f oracle %.9f" % (tag, it, float(res['loss']), float(tot)))
        sd_ref = m.state_dict()
        print("[frozen %s] max param dev oracle vs ref %.3e; running_mean dev %.3e; trainable %d of %d" % (
            tag, max(dev(tr.sd[k], sd_ref[k]) for k in tr.names),
            max(dev(tr.sd[k], sd_ref[k]) for k in sd_ref if k.endswith('running_mean')),
            sum(p.requires_grad for p in m.parameters()), len(list(m.parameters()))))
        out[tag + "_rm_final"] = npy(torch.cat([sd_ref[k].flatten() for k in sd_ref if k.endswith('running_mean')]))
        out[tag + "_rv_final"] = npy(torch.cat([sd_ref[k].flatten() for k in sd_ref if k.endswith('running_var')]))
        out[tag + "_nbt_final"] = npy(torch.stack([sd_ref[k] for k in sd_ref if k.endswith('num_batches_tracked')]))
    np.savez_compressed(os.path.join(GOLD, "frozen.npz"), **out)


def gen_concat():
    """vision_base ConcatDataset over three tiny children: which (child, local index) every global index maps to,
    and how common keywords / per-child overrides reach the children"""
    import json
    from vision_base.data.datasets.dataset_utils import ConcatDataset
    cfgs = [dict(name="tiny_ds.Tiny", n=5, tag="a"), dict(name="tiny_ds.Tiny", n=3, tag="b", scale=10),
            dict(name="tiny_ds.Tiny", n=7, tag="c")]
    ds = ConcatDataset([EasyDict(c) for c in cfgs], scale=2)
    rec = dict(cfgs=cfgs, common=dict(scale=2), length=int(len(ds)), items=[{k: (int(v) if k == "value" else v) for k, v in ds[i].items()} for i in range(len(ds))])
    json.dump(rec, open(os.path.join(GOLD, "concat_dataset.json"), "w"))
    print("concat dataset: %d items" % len(ds))


def gen_velo_gt():
    """KITTI ground-truth export (monodepth_utils.py:368-420 generate_depth_map, with its sub2ind quirk) over the seeded
    on-disk tree of tests/helpers_kitti.py: the reference's depth maps for the split's frames"""
    import tempfile
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers_kitti as HK
    from monodepth.networks.utils.monodepth_utils import generate_depth_map
    with tempfile.TemporaryDirectory() as d:
        raw, split = HK.make_tree(d)
        HK.add_velodyne(raw)
        gts = []
        for line in open(split):
            folder, frame_id, _ = line.split()
            gts.append(generate_depth_map(os.path.join(raw, folder.split("/")[0]),
                                          os.path.join(raw, folder, "velodyne_points/data", "%010d.bin" % int(frame_id)), 2, True))
    gts = np.array(gts)
    print("velodyne gt: %s, %.1f %% of the pixels hit, depth range %.2f..%.2f" % (
        gts.shape, 100 * float((gts > 0).mean()), float(gts[gts > 0].min()), float(gts.max())))
    np.savez_compressed(os.path.join(GOLD, "velo_gt.npz"), gt=gts)


def gen_teacher_keys():
    """monodepth/transform_teacher.py on a checkpoint of the reference depth+pose meta-arch: the key list of the
    teacher state_dict (order included) and a checksum per kept tensor."""
    import json
    import tempfile
    from monodepth.transform_teacher import transform_teacher_model
    m = ref_model(64, 128, True)
    m.load_state_dict({k: v.clone() for k, v in O.init_state(seed=1, with_pose=True).items()}, strict=True)
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "stage1.pth"), os.path.join(d, "teacher.pth")
        torch.save({"model_state_dict": m.state_dict(), "optimizer_state_dict": {}}, src)
        transform_teacher_model(src, dst)
        out = torch.load(dst, map_location="cpu")
    rec = {"src_keys": list(m.state_dict().keys()), "dst_keys": list(out.keys()),
           "dst_sums": [float(v.double().sum()) for v in out.values()], "is_bare_state_dict": isinstance(out, dict)
           and "model_state_dict" not in out}
    json.dump(rec, open(os.path.join(GOLD, "teacher_keys.json"), "w"))
    print("teacher keys: %d of %d kept" % (len(rec["dst_keys"]), len(rec["src_keys"])))


def gen_onnx():
    """scripts/onnx_export.py:39-52 on the real MonoDepthWPose (CPU, eval mode): `dummy_forward(image)` and the
    reference's own `torch.onnx.export(..., opset_version=11)` of it — operator histogram, graph input / output
    signature and initializer shapes of the file the reference writes (read back with the wire-format reader; the
    `onnx` package is absent, so the exporter's onnxscript splice is skipped — see fsnet_amd/export/onnx_graph.py)."""
    import collections
    import io
    import json
    import warnings
    from fsnet_amd.export import onnx_graph as G
    from tests.helpers_onnx import case as onnx_case
    sd0, image = onnx_case()
    m = ref_model(64, 128, False)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m.eval()
    with torch.no_grad():
        pred = m.dummy_forward(image)
    assert list(pred.keys()) == ["depth"]
    m.forward = m.dummy_forward
    f = io.BytesIO()
    with warnings.catch_warnings(), G._without_onnx_package():
        warnings.simplefilter("ignore")
        torch.onnx.export(m, torch.zeros(1, 3, 64, 128), f, input_names=['input'], output_names=['output'],
                          opset_version=11, dynamo=False)
    model = G.read_model(f)
    G.check_model(model)
    g = model["graph"]
    rec = {"ir_version": model["ir_version"], "opsets": model["opsets"],
           "ops": dict(collections.Counter(n["op_type"] for n in g["nodes"])),
           "inputs": [v for v in g["inputs"] if v["name"] not in g["initializers"]],
           "outputs": [dict(name=v["name"], elem_type=v["elem_type"], rank=len(v["shape"])) for v in g["outputs"]],
           "initializer_shapes": sorted([list(a.shape) for a in g["initializers"].values()]),
           "n_bytes": len(f.getvalue())}
    json.dump(rec, open(os.path.join(GOLD, "onnx_graph.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "onnx_dummy_forward.npz"), depth=npy(pred["depth"]))
    print("onnx: %d nodes, %d initializers, depth range %.3f..%.3f" % (
        len(g["nodes"]), len(g["initializers"]), float(pred["depth"].min()), float(pred["depth"].max())))


if __name__ == "__main__":
    if "--only-onnx" in sys.argv:
        gen_onnx()
        sys.exit(0)
    if "--only-velo" in sys.argv:
        gen_velo_gt()
        sys.exit(0)
    if "--only-noov" in sys.argv:
        gen_no_overlap_mask()
        sys.exit(0)
    if "--only-concat" in sys.argv:
        gen_concat()
        sys.exit(0)
    if "--only-frozen" in sys.argv:
        gen_frozen()
        sys.exit(0)
    if "--only-sigmoid" in sys.argv:
        gen_sigmoid_decoder()
        sys.exit(0)
    if "--only-teacher" in sys.argv:
        gen_teacher_keys()
        sys.exit(0)
    if "--only-augment" in sys.argv:
        gen_augment()
        sys.exit(0)
    if "--only-augment-resize" in sys.argv:
        gen_augment_resize()
        sys.exit(0)
    if "--only-fisheye" in sys.argv:
        gen_fisheye()
        sys.exit(0)
    if "--only-options" in sys.argv:
        gen_loss_options()
        sys.exit(0)
    if "--only-kitti" in sys.argv:
        gen_kitti_dataset()
        sys.exit(0)
    if "--only-r50fx" in sys.argv:
        gen_model_r50fx()
        sys.exit(0)
    gen_ops()
    gen_loss_chain()
    gen_model(True, "depthpose")
    gen_model(False, "wpose")
    gen_eval()
    gen_distill()
    gen_augment()
    gen_fisheye()
    gen_model_r50fx()
    gen_kitti_dataset()
    gen_loss_options()
    gen_teacher_keys()
    gen_sigmoid_decoder()
    gen_frozen()
    gen_concat()
    gen_augment_resize()
    gen_no_overlap_mask()
    gen_velo_gt()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")
