"""How long does the HOST spend inside graph.replay() per step, versus the step time on the device?  (If the two are
close the step is submission-bound: branches of the graph reach the GPU in the runtime's submission order.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build

dev = torch.device("cuda", 0)
RT.set_compute_dtype("bf16")
model = build(**meta_arch_cfg(192, 640, with_pose=True, depth=18)).to(dev).train()
tc = training_cfg(clip_gradients=35.0, lr=1e-4)
opt = build_optimizer(model, **tc.optimizer)
hook = build(**tc.training_hook)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
for i in range(10):
    hook(dict(batches[i % 4]), model, opt, global_step=i)
torch.cuda.synchronize()
g = hook._g["graph"]
host, total = [], []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print("graph.replay() host time: median %.3f ms (min %.3f); launch -> device idle: median %.3f ms (min %.3f)"
      % (host[15], host[0], total[15], total[0]))
# back-to-back replays (the benchmark's regime)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("50 back-to-back replays: host %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) * 20, (t2 - t0) * 20))
