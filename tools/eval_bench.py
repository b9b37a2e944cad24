"""Throughput of the validation path on one MI355X: eval-mode forward of the depth network at
192x640 + inverse-depth resize to 375x1242 + device-side metrics, against the oracle's numpy metrics on the host.
    python tools/eval_bench.py > profiles/<name>.json"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from fsnet_amd.configs import meta_arch_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.hip import ops
from fsnet_amd.vision_base.utils.builder import build
from oracle import eval_oracle as EO
from oracle import fsnet_oracle as O

dev = torch.device("cuda", 0)
RT.set_compute_dtype("bf16")
B, h, w, H, W = 16, 192, 640, 375, 1242
m = build(**meta_arch_cfg(h, w, with_pose=False)).to(dev).eval()
data = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(B, h, w, seed=0).items()}
rng = np.random.RandomState(0)
gt_np = np.zeros((B, H, W), np.float32)
msk = rng.rand(B, H, W) < 0.05
gt_np[msk] = (rng.rand(int(msk.sum())) * 79 + 1).astype(np.float32)
gt = torch.from_numpy(gt_np).to(dev)


def step():
    with torch.no_grad():
        depth = m(data, dict(is_training=False))["depth"][:, 0].float()
        full = torch.stack([ops.resize_linear(depth[i], H, W, invert=True) for i in range(B)])
        return ops.depth_eval(full, gt)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 20
for _ in range(n):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
full = torch.rand(B, H, W, device=dev) * 40 + 1
s.record()
for _ in range(20):
    ops.depth_eval(full, gt)
e.record(); torch.cuda.synchronize()
k_us = s.elapsed_time(e) / 20 * 1e3
t1 = time.perf_counter()
d0 = full[0].cpu().numpy()
for _ in range(3):
    EO.single_loss(d0.copy(), gt_np[0].copy())
cpu = (time.perf_counter() - t1) / 3
print(json.dumps({"metric": "validation images/s (eval forward 192x640 bf16 + resize to 375x1242 + depth metrics)",
                  "value": round(B / dt, 1), "batch": B, "ms_per_batch": round(dt * 1e3, 3),
                  "depth_eval_kernel_us_per_batch": round(k_us, 1),
                  "depth_eval_algorithmic_GBps": round(B * H * W * 8 / (k_us * 1e-6) / 1e9, 1),
                  "cpu_metrics_images_per_s_numpy_oracle": round(1 / cpu, 1), "n_valid_0": float(out[0, 15])}))
