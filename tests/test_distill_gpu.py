"""Self-distillation stage on the HIP path: DistillWPoseMeta built through the registry, against
the golden vectors of the REAL reference (tests/golden/distill.npz) and the CPU oracle's per-parameter gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import distill_oracle as D
from oracle import fsnet_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = "fsnet_amd."


def _cfg(H, W, teacher_path=None):
    from easydict import EasyDict as edict
    enc = dict(name=P + 'vision_base.networks.models.backbone.resnet.resnet', depth=18, pretrained=False, frozen_stages=-1,
               num_stages=4, out_indices=(-1, 0, 1, 2, 3), norm_eval=False, dilations=(1, 1, 1, 1))
    dec = dict(num_ch_enc=np.array([64, 64, 128, 256, 512]), num_output_channels=16, use_skips=True, scales=[0, 1, 2, 3],
               min_depth=0.5, max_depth=100)
    return edict(
        name=P + 'monodepth.networks.models.meta_archs.monodepth2_model.DistillWPoseMeta',
        teacher_net_cfg=edict(name=P + 'monodepth.networks.models.meta_archs.teacher_model.MonoDepthInference',
                              backbone_cfg=edict(**enc),
                              depth_head_cfg=edict(name=P + 'monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder', **dec)),
        teacher_net_path=teacher_path,
        depth_backbone_cfg=edict(**enc),
        head_cfg=edict(name=P + 'monodepth.networks.models.heads.monodepth2_decoder.MonoDepth2Decoder',
                       scales=[0, 1, 2, 3], height=H, width=W, min_depth=0.5, max_depth=100.0, overlapped_mask=True,
                       is_log_image=False, distillation_loss_weight=0.3, is_uncertain_distill=True,
                       depth_decoder_cfg=edict(name=P + 'monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoderUncertain', **dec)),
        train_cfg=edict(frame_ids=[0, 1, -1]), test_cfg=edict())


def test_state_dict_names_match_the_reference_layout():
    from fsnet_amd.vision_base.utils.builder import build
    m = build(**_cfg(64, 128))
    want = D.init_states()
    got = m.state_dict()
    assert set(got) == set(want)
    assert all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)


def test_fp32_step_matches_reference_golden_and_oracle_gradients(dev, tmp_path):
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.float32)
    RT.tie_noise = False
    g = np.load(os.path.join(GOLD, "distill.npz"))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    sd = D.init_states(seed=int(g["seed"]), teacher_seed=int(g["teacher_seed"]))
    tpath = str(tmp_path / "teacher.pth")            # the reference loads the teacher from a checkpoint file
    torch.save({k[len("teacher_net."):]: v.clone() for k, v in sd.items() if k.startswith("teacher_net.")}, tpath)
    m = build(**_cfg(H, W, tpath))
    student = {k: v.clone() for k, v in sd.items() if not k.startswith("teacher_net.")}
    missing = m.load_state_dict(student, strict=False)
    assert all(k.startswith("teacher_net.") for k in missing.missing_keys) and not missing.unexpected_keys
    m = m.to(dev).train()
    assert not m.teacher_net.training and not any(p.requires_grad for p in m.teacher_net.parameters())
    data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(B, H, W, seed=int(g["batch_seed"])).items()}
    out = m(data, dict(is_training=True))
    out["loss"].backward()
    torch.cuda.synchronize()
    loss = float(out["loss"].detach())
    assert abs(loss - float(g["loss"])) < 3e-4 * abs(float(g["loss"])), (loss, float(g["loss"]))
    for s in range(4):
        got = float(out["loss_dict"]["distilation/%d" % s])
        assert abs(got - float(g["ld_distilation_%d" % s])) < 3e-4 * float(g["ld_distilation_%d" % s]), (s, got)
    teacher0 = m.teacher_net.compute_teacher_depth(data[("image", 0)])[("teacher_depth", 0, 0)]
    assert abs(float(teacher0.mean()) - float(g["teacher_depth_0_mean"])) < 1e-4 * float(g["teacher_depth_0_mean"])
    # gradient norms of every student parameter against the reference
    names = [k for k, _ in m.named_parameters() if not k.startswith("teacher_net.")]
    gn = torch.stack([dict(m.named_parameters())[k].grad.norm() for k in names]).cpu().numpy()
    ref = g["gradnorm"]
    big = ref > 1e-3 * ref.max()
    assert np.abs(gn - ref)[big].max() / ref[big].max() < 2e-2, float(np.abs(gn - ref)[big].max() / ref[big].max())
    gu = dict(m.named_parameters())["head.depth_decoder.decoder.14.weight"].grad.cpu().numpy()
    assert np.abs(gu - g["unc_w_grad"]).max() < 2e-3 * np.abs(g["unc_w_grad"]).max()
    # teacher parameters receive nothing
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in m.teacher_net.parameters())


def test_distill_kernels_against_torch(dev):
    from fsnet_amd.hip import ops
    gen = torch.Generator().manual_seed(0)
    p = (torch.rand(3, 1, 24, 40, generator=gen) * 50 + 1).to(dev)
    t = (torch.rand(3, 1, 24, 40, generator=gen) * 50 + 1).to(dev)
    u = (torch.rand(3, 1, 24, 40, generator=gen) * 0.9 + 0.05).to(dev)
    for unc in (None, u):
        pr, ur = p.clone().requires_grad_(True), (None if unc is None else unc.clone().requires_grad_(True))
        ref = D.distill_loss(pr, t, ur)
        (ref * 1.7).backward()
        got = ops.distill_fwd(p, t, unc)
        assert abs(float(got) - float(ref.detach())) < 1e-6 * abs(float(ref.detach()))
        dp, du = ops.distill_bwd(p, t, unc, torch.tensor(1.7, dtype=torch.float64, device=dev))
        assert float((dp - pr.grad).abs().max()) < 1e-6 * float(pr.grad.abs().max())
        if unc is not None:
            assert float((du - ur.grad).abs().max()) < 1e-5 * float(ur.grad.abs().max())
    logits = torch.randn(2, 8, 12, 16, generator=gen).to(dev)
    s = ops.sigmoid_head_fwd(logits)
    assert float((s[:, 0] - torch.sigmoid(logits[..., 0])).abs().max()) < 1e-6
    du = torch.randn(2, 1, 8, 12, generator=gen).to(dev)
    dl = ops.sigmoid_head_bwd(s, du, 16, torch.float32)
    want = du[:, 0] * s[:, 0] * (1 - s[:, 0])
    assert float((dl[..., 0] - want).abs().max()) < 1e-6 and float(dl[..., 1:].abs().max()) == 0


def test_distill_training_steps_through_the_hook(dev):
    """bf16, three optimisation steps through BaseTrainingHook (eager + hipGraph): finite, loss moves, teacher frozen"""
    from fsnet_amd.configs import training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.bfloat16)
    sd = D.init_states(seed=5, teacher_seed=6)
    m = build(**_cfg(64, 128))
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    t0 = torch.cat([p.detach().flatten().clone() for p in m.teacher_net.parameters()])
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(graph_warmup=2, **tc.training_hook)
    losses = []
    for it in range(5):
        out = hook(dict(O.synthetic_batch(2, 64, 128, seed=9)), m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e4 for l in losses) and losses[-1] < losses[0], losses
    assert hook.graph_captures == 1 and hook.graph_replays == 2
    t1 = torch.cat([p.detach().flatten() for p in m.teacher_net.parameters()])
    assert float((t0 - t1).abs().max()) == 0.0
    RT.set_compute_dtype(torch.float32)
