"""Does bf16 TRAIN like fp32?    The reference computes in fp32 only (no autocast / half
anywhere); this build's headline lines are bf16.  Gradient cosines at random initialisation (tests/test_model_gpu.py) and
loss curves on the throughput batches — shifted noise-like fields whose loss cannot fall — do not answer the question.
Here the networks are trained on a scene that CAN be learned: tests/helpers_scene.py renders three frames of one rigid
corridor by ray casting, tests/test_scene_cpu.py pins (against the oracle's loss chain) that the generating depth and
motion are the photometric minimum.  From the same weights, on the same batches, fp32 / bf16 / fp32 again:

  * the loss falls by more than 30 % in every run (measured 55-65 %),
  * the bf16 run ends where the fp32 runs end (within 3x the spread of the two fp32 runs, or 10 %),
  * the predicted depth of held-out scenes follows the generating depth (log-depth correlation > 0.8, measured 0.93-0.97).

MonoDepthWPose (dataset poses: configs/kitti_wpose_example, multi_dataset_example — the shipped ResNet-50 configuration) for
ResNet-18 and ResNet-50.  MonoDepthMeta with a pose network that starts from RANDOM weights is not asserted on: on this loss
(monodepth2_decoder.py:248-262, the identity auto-mask) such a run either learns or — when an early update makes every
reprojection worse than the identity term — is left without a gradient for good, and which of the two happens is decided by
the seed, for fp32 and bf16 alike (profiles/r06_scene_training.txt: 14 runs; the reference ships ImageNet-pretrained encoders).
"""
import numpy as np
import pytest
import torch

from oracle import fsnet_oracle as O
from tests.helpers_scene import corridor_batch, log_depth_correlation

gpu = pytest.mark.gpu
H, W, B, NB = 96, 320, 4, 64


def _train(dev, depth, dtype, sd0, pool, val, steps):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(dtype)
    RT.tie_noise = True
    m = build(**meta_arch_cfg(H, W, with_pose=False, depth=depth))
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m = m.to(dev).train()
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(**tc.training_hook)                      # clip 35, Adam 1e-4, hipGraph replay: the benchmarked step
    losses = []
    for it in range(steps):
        out = hook(dict(pool[it % NB][0]), m, opt)
        losses.append(out["loss"].detach().clone())
    torch.cuda.synchronize()
    assert hook.graph_replays >= steps - 8
    L = torch.stack(losses).double().cpu().numpy()
    m.eval()
    cors, ratios = [], []
    with torch.no_grad():
        for b, t in val:
            d = m(dict(b), dict(is_training=False))["depth"]
            cors.append(log_depth_correlation(d, t["depth"]))
            ratios.append(float((d / t["depth"]).median()))
    return L, float(np.mean(cors)), float(np.median(ratios))


@gpu
@pytest.mark.parametrize("depth,steps", [(18, 400), (50, 600)], ids=["resnet18", "resnet50"])
def test_bf16_trains_like_fp32_on_a_learnable_scene(dev, depth, steps):
    from fsnet_amd.engine.runtime import RT
    sd0 = O.init_state(seed=21, depth=depth, with_pose=False)
    pool = [corridor_batch(B, H, W, seed=4000 + i, device=dev) for i in range(NB)]
    val = [corridor_batch(B, H, W, seed=9000 + i, device=dev) for i in range(4)]
    try:
        runs = {name: _train(dev, depth, dt, sd0, pool, val, steps)
                for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16), ("fp32 again", torch.float32))}
    finally:
        RT.set_compute_dtype(torch.bfloat16)
    head, tail = max(1, steps // 20), max(1, steps // 10)
    first = {k: float(r[0][:head].mean()) for k, r in runs.items()}
    last = {k: float(r[0][-tail:].mean()) for k, r in runs.items()}
    for k, (L, corr, ratio) in runs.items():
        print("ResNet-%d %-10s loss %.4f -> %.4f (%.0f %% down), held-out log-depth correlation %.3f, median depth ratio %.2f"
              % (depth, k, first[k], last[k], 100 * (1 - last[k] / first[k]), corr, ratio))
        assert np.isfinite(L).all()
        assert last[k] < 0.7 * first[k], (k, first[k], last[k])
        assert corr > 0.8, (k, corr)
        assert 0.6 < ratio < 1.6, (k, ratio)             # dataset poses fix the scale: the depth is metric
    spread = abs(last["fp32"] - last["fp32 again"])
    assert abs(last["bf16"] - last["fp32"]) < max(3.0 * spread, 0.10 * last["fp32"]), (last, spread)
    # the same start for everybody (the first step's loss is the untrained network's)
    assert abs(runs["bf16"][0][0] - runs["fp32"][0][0]) < 0.02 * runs["fp32"][0][0]
