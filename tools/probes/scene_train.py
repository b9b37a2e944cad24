"""Exploration for tests/test_trains_gpu.py: R18 / R50 depth+pose on the corridor scene, fp32 and bf16 from the same weights.
    python tools/probes/scene_train.py [depth=18] [steps=300] [H=96] [W=320] [B=4] [lr=1e-4] [pose=1]"""
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
depth, steps = int(kw.get("depth", 18)), int(kw.get("steps", 300))
H, W, B, lr, pose = int(kw.get("H", 96)), int(kw.get("W", 320)), int(kw.get("B", 4)), float(kw.get("lr", 1e-4)), kw.get("pose", "1") == "1"
fixed = kw.get("fixed", "0") == "1"
NB = int(kw.get("nb", 64))
dev = torch.device("cuda", 0)
sd0 = O.init_state(seed=int(kw.get('seed', 21)), depth=depth, with_pose=pose)
pool = [corridor_batch(B, H, W, seed=4000 + i, device=dev, fixed_geometry=fixed, scale=float(kw.get('scale', 1.0))) for i in range(NB)]
val = [corridor_batch(B, H, W, seed=9000 + i, device=dev, fixed_geometry=fixed, scale=float(kw.get('scale', 1.0))) for i in range(4)]
if kw.get("mm", "0") == "1":      # the reference's precomputed-motion-mask branch with an all-zero mask: no identity competitor
    for b, _ in pool:
        b["motion_mask"] = torch.zeros(B, H, W, device=dev)
seed0 = int(kw.get("seed", 21))
for dtype in (kw.get("dtypes", "fp32,bf16,fp32")).split(","):
    RT.set_compute_dtype(dtype)
    RT.tie_noise = True
    m = build(**meta_arch_cfg(H, W, with_pose=pose, depth=depth))
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m = m.to(dev).train()
    tc = training_cfg(lr=lr)
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(**tc.training_hook)
    t0 = time.time()
    losses = []
    for it in range(steps):
        out = hook(dict(pool[it % NB][0]), m, opt)
        losses.append(out["loss"].detach().clone())
    torch.cuda.synchronize()
    L = torch.stack(losses).double().cpu()
    m.eval()
    cors = []
    with torch.no_grad():
        for b, t in val:
            d = m(dict(b), dict(is_training=False))["depth"]
            cors.append(log_depth_correlation(d, t["depth"]))
    w = max(1, steps // 10)
    print("R%d %s pose=%s: loss first %.4f  ... %s ... last %.4f ; val log-depth corr %.3f ; %.1f s" % (
        depth, dtype, pose, float(L[:w].mean()), " ".join("%.4f" % float(L[i:i + w].mean()) for i in range(w, steps - w, w)),
        float(L[-w:].mean()), sum(cors) / len(cors), time.time() - t0), flush=True)
