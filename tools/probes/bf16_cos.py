"""per-parameter cosine of the bf16 step's convolution gradients against the fp32 CPU oracle, for a depth / size given on the
command line: python tools/probes/bf16_cos.py depth H W B [with_pose]  (environment switches apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fsnet_oracle as O
from fsnet_amd.configs import meta_arch_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.utils.builder import build

depth, H, W, B = (int(v) for v in sys.argv[1:5])
with_pose = len(sys.argv) > 5 and sys.argv[5] == "1"
dt = os.environ.get("PROBE_DTYPE", "bf16")
dev = torch.device("cuda:0")
RT.set_compute_dtype(dt)
RT.tie_noise = False
sd0 = O.init_state(seed=11, depth=depth, with_pose=with_pose)
m = build(**meta_arch_cfg(H, W, with_pose=with_pose, depth=depth))
m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
m = m.to(dev).train()
data = O.synthetic_batch(B, H, W, seed=13)
out = m({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}, dict(is_training=True))
out["loss"].backward()
torch.cuda.synchronize()
tr = O.OracleTrainer(sd0, depth=depth, with_pose=with_pose, clip=None)
total, ld, _, raw, _ = tr.step(data)
gmax = max(float(r.norm()) for r in raw.values())
groups = {}
for k, p in m.named_parameters():
    ref = raw[k]
    if float(ref.norm()) < 1e-3 * gmax or ref.dim() != 4:
        continue
    g = p.grad.cpu()
    cos = float((g * ref).sum() / (g.norm() * ref.norm()))
    key = ".".join(k.split(".")[:2]) if "backbone" in k else "decoder"
    groups.setdefault(key, []).append(cos)
print("depth %d %dx%d B=%d %s loss %.6f (oracle %.6f)" % (depth, H, W, B, dt, float(out["loss"]), float(total)))
for k, v in groups.items():
    print("   %-32s n=%2d  min %.3f  mean %.3f" % (k, len(v), min(v), sum(v) / len(v)))
