"""Exploration for tests/test_trains_gpu.py, learned pose: the depth network first trained with the dataset's poses
(MonoDepthWPose), then MonoDepthMeta (depth + pose networks, pose net from random weights) from those depth weights.
    python tools/probes/scene_train2.py [depth=18] [pre=400] [steps=800] [seed=21]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build
from oracle import fsnet_oracle as O
from tests.helpers_scene import corridor_batch, log_depth_correlation

kw = dict(a.split("=") for a in sys.argv[1:])
depth, pre, steps, seed = int(kw.get("depth", 18)), int(kw.get("pre", 400)), int(kw.get("steps", 800)), int(kw.get("seed", 21))
H, W, B = int(kw.get("H", 96)), int(kw.get("W", 320)), int(kw.get("B", 4))
NB = 64
dev = torch.device("cuda", 0)
SC = float(kw.get('scale', 0.1))
pool = [corridor_batch(B, H, W, seed=4000 + i, device=dev, scale=SC) for i in range(NB)]
val = [corridor_batch(B, H, W, seed=9000 + i, device=dev, scale=SC) for i in range(4)]


def train(m, n, dtype, start=0):
    RT.set_compute_dtype(dtype)
    RT.tie_noise = True
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(**tc.training_hook)
    losses = []
    for it in range(n):
        out = hook(dict(pool[(start + it) % NB][0]), m, opt)
        losses.append(out["loss"].detach().clone())
    torch.cuda.synchronize()
    L = torch.stack(losses).double().cpu()
    m.eval()
    cors = []
    with torch.no_grad():
        for b, t in val:
            cors.append(log_depth_correlation(m(dict(b), dict(is_training=False))["depth"], t["depth"]))
    m.train()
    return L, sum(cors) / len(cors)


def fmt(L):
    w = max(1, len(L) // 10)
    return " ".join("%.4f" % float(L[i:i + w].mean()) for i in range(0, len(L) - w + 1, w))


sd_w = O.init_state(seed=seed, depth=depth, with_pose=False)
m = build(**meta_arch_cfg(H, W, with_pose=False, depth=depth))
m.load_state_dict({k: v.clone() for k, v in sd_w.items()}, strict=True)
m = m.to(dev).train()
L, c = train(m, pre, "fp32")
print("R%d WPose fp32 %d steps: %s ; corr %.3f" % (depth, pre, fmt(L), c), flush=True)
trained = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
sd_p = O.init_state(seed=seed + 1, depth=depth, with_pose=True)
for k in sd_p:
    if k in trained and not k.startswith("pose_") and "pose_decoder" not in k:
        sd_p[k] = trained[k]
for dtype in ("fp32", "bf16", "fp32"):
    m2 = build(**meta_arch_cfg(H, W, with_pose=True, depth=depth))
    m2.load_state_dict({k: v.clone() for k, v in sd_p.items()}, strict=True)
    m2 = m2.to(dev).train()
    t0 = time.time()
    L, c = train(m2, steps, dtype, start=pre)
    print("R%d depth+pose %s %d steps: %s ; corr %.3f ; %.1f s" % (depth, dtype, steps, fmt(L), c, time.time() - t0), flush=True)
