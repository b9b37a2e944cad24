"""hipGraph replay of the training step == eager execution of the same steps (same seeds, same device-side
tie-break noise sequence, lr schedule changes picked up through the device-resident lr scalar)."""
import pytest
import torch

from oracle import fsnet_oracle as O

pytestmark = pytest.mark.gpu


def _run(dev, use_graph, steps, dtype, tie_noise):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(dtype)
    RT.tie_noise = tie_noise
    B, H, W = 2, 64, 128
    m = build(**meta_arch_cfg(H, W, with_pose=True))
    m.load_state_dict(O.init_state(seed=3, with_pose=True), strict=True)
    m = m.to(dev).train()
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(use_graph=use_graph, graph_warmup=2, **tc.training_hook)
    losses = []
    for it in range(steps):
        if it == 4:
            opt.param_groups[0]["lr"] *= 0.5          # what a scheduler does between steps
        out = hook(dict(O.synthetic_batch(B, H, W, seed=50 + it)), m, opt)
        losses.append(out["loss"].detach().clone())
    torch.cuda.synchronize()
    params = torch.cat([p.detach().double().flatten() for p in m.parameters()]).cpu()
    rm = torch.cat([v.double().flatten() for k, v in m.state_dict().items() if "running_" in k]).cpu()
    if tie_noise:
        assert int(m.head._pl.seed_buf.item()) == steps       # device-side seed stream advanced once per step
    RT.tie_noise = False
    return torch.stack(losses).cpu(), params, rm, hook, opt


@pytest.mark.parametrize("dtype,tie", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, True)])
def test_graph_replay_matches_eager(dev, dtype, tie):
    le, pe, re_, he, oe = _run(dev, False, 6, dtype, tie)
    l2, p2, r2, _, _ = _run(dev, False, 6, dtype, tie)
    lg, pg, rg, hg, og = _run(dev, True, 6, dtype, tie)
    assert he.graph_replays == 0 and hg.graph_replays == 3          # steps 0,1 eager; 2 captured; 3..5 replayed
    # same kernels in the same launch order; only the order of fp atomics differs run to run, and min-selection /
    # ReLU flips amplify that over steps.  Yardstick: the spread between two EAGER runs of the same thing.
    def d(a, b, rel_floor=0.0):
        return float(((a - b).abs() / a.abs().clamp_min(rel_floor)).max()) if rel_floor else float((a - b).abs().max())
    # Measured with tools/probes/graph_vs_eager.py (3 eager and 3 graph runs each): the spread is bimodal — runs agree
    # to ~1e-5 (loss) / 5e-4 (running statistics) or, when one near-tie min-selection / ReLU decision flips on a
    # reordered fp32 atomic, differ by ~8e-5 / 5e-3 — eager against eager just like graph against eager.
    f32 = dtype == torch.float32
    # (floors with a 3x margin over the largest spreads seen in ~20 suite runs: these are chaotic trajectories)
    assert d(le, lg, 1e-9) < max(1e-3 if f32 else 1e-2, 5 * d(le, l2, 1e-9)), (le, lg, l2)
    assert d(pe, pg) < max(2e-3 if f32 else 4e-3, 5 * d(pe, p2)), (d(pe, pg), d(pe, p2))
    assert d(re_, rg, 1.0) < max(3e-2 if f32 else 5e-2, 5 * d(re_, r2, 1.0)), (d(re_, rg, 1.0), d(re_, r2, 1.0))
    assert oe._step_count_fused == og._step_count_fused == 6
    assert int(og._step_buf.item()) == 6 and abs(float(og._lr_buf.item()) - og.param_groups[0]["lr"]) < 1e-9


def test_graph_falls_back_on_new_batch_shape(dev):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.bfloat16)
    m = build(**meta_arch_cfg(64, 128, with_pose=True)).to(dev).train()
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(graph_warmup=2, **tc.training_hook)
    for it in range(4):
        hook(dict(O.synthetic_batch(2, 64, 128, seed=it)), m, opt)
    assert hook.graph_replays == 1 and hook.graph_captures == 1
    out = hook(dict(O.synthetic_batch(1, 64, 128, seed=9)), m, opt)     # last, ragged batch of an epoch
    assert hook.graph_replays == 1 and torch.isfinite(out["loss"])
    for it in range(4):
        out = hook(dict(O.synthetic_batch(2, 64, 128, seed=20 + it)), m, opt)
    torch.cuda.synchronize()
    assert hook.graph_replays == 2 and hook.graph_captures == 2 and hook.use_graph
    assert torch.isfinite(out["loss"]) and opt._step_count_fused == 9


@pytest.mark.parametrize("arch", ["two equal chains", "resnet34 + resnet18", "one chain"])
def test_replayed_step_leaves_every_parameter_gradient(dev, arch):
    """Inside a capture the weight-gradient batches are issued late, and with two equal chains partly on the pose chain's
    streams (nets.flush_deferred): every parameter's gradient after the first replayed step is the eager step's (a batch
    dropped or issued against the wrong event leaves zeros or garbage in ITS layers only — relative error 1 or more — and
    the loss and parameter comparisons above have tolerances a few layers can hide in.  Two EAGER runs of this step differ
    by up to 4e-2 in single encoder layers, tools/probes/grad_late.py: per-pixel minima that flip with the summation order
    of the step before)."""
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine import nets
    from tests.helpers_scene import corridor_batch
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.float32)
    B, H, W = 2, 64, 128
    with_pose = arch != "one chain"
    depth = 34 if arch.startswith("resnet34") else 18
    cfg = meta_arch_cfg(H, W, with_pose=with_pose, depth=depth)
    if depth == 34:
        cfg["pose_backbone_cfg"]["depth"] = 18
    sd0 = {k: v.clone() for k, v in build(**cfg).state_dict().items()}
    grads = {}
    try:
        for use_graph in (False, True):
            m = build(**cfg)
            m.load_state_dict(sd0, strict=True)
            m = m.to(dev).train()
            tc = training_cfg()
            opt = build_optimizer(m, **tc.optimizer)
            hook = build(use_graph=use_graph, graph_warmup=2, **tc.training_hook)
            before = dict(nets.HANDOVERS)
            for it in range(3):                        # two eager steps, then the captured step's first replay
                hook(dict(corridor_batch(B, H, W, seed=90 + it, device=dev)[0]), m, opt)
            torch.cuda.synchronize()
            assert hook.graph_captures == (1 if use_graph else 0)
            grads[use_graph] = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
            if use_graph:
                assert nets.HANDOVERS["late"] > before["late"]
                assert (nets.HANDOVERS["shared"] > before["shared"]) == (arch == "two equal chains")
                assert not nets._LATE and not nets._AT_END
    finally:
        RT.set_compute_dtype(torch.bfloat16)
    gmax = max(float(g.norm()) for g in grads[False].values())
    assert set(grads[True]) == set(grads[False])
    for k, ge in grads[False].items():
        if float(ge.norm()) < 1e-6 * gmax:            # (a convolution bias in front of a BatchNorm: zero up to rounding)
            continue
        rel = float((grads[True][k] - ge).norm() / ge.norm())
        assert rel < 0.15, (arch, k, rel)
