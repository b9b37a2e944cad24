"""Which torch-native (aten) kernels does one training step launch, and from where?  (eager step, torch.profiler)"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build

dev = torch.device("cuda", 0)
RT.set_compute_dtype("bf16")
model = build(**meta_arch_cfg(192, 640, with_pose=True)).to(dev).train()
tc = training_cfg(clip_gradients=35.0, lr=1e-4)
opt = build_optimizer(model, **tc.optimizer)
hook = build(use_graph=False, **tc.training_hook)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
for i in range(3):
    hook(dict(batches[i % len(batches)]), model, opt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    hook(dict(batches[0]), model, opt)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue      # count the leaf op that owns the kernel
    where = "?"
    for fr in (ev.stack or []):
        if "fsnet_amd" in fr and "site-packages" not in fr:
            where = fr.split("fsnet_amd/")[-1]
            break
    cnt[(ev.name, str(ev.input_shapes)[:60], where)] += 1
tot = 0
for (name, shp, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d  %-22s %-62s %s" % (c, name, shp, where))
    tot += c
print("total aten kernels per step:", tot)
