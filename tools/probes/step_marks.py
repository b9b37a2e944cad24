"""Unprofiled timeline of a replayed step: device-clock marks (FSNET_AMD_MARKS=1) stamped on each chain's stream.
    python tools/probes/step_marks.py [dp]      dp: the data-parallel step at world size 1 over RCCL (set FSNET_AMD_LANES)"""
import os, sys, time
os.environ["FSNET_AMD_MARKS"] = "1"
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
if "dp" in sys.argv:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from fsnet_amd.engine.dataparallel import DataParallelContext
    dist.init_process_group("nccl", rank=0, world_size=1)
    hook.tune_steps = 0
    RT.dp = DataParallelContext(model)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
for i in range(12):
    hook(dict(batches[i % 4]), model, opt, global_step=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    hook(dict(batches[i % 4]), model, opt, global_step=12 + i)
torch.cuda.synchronize()
print("ms/step with marks: %.3f (replays %d)" % ((time.perf_counter() - t0) * 20, hook.graph_replays))
for rep in range(2):
    hook(dict(batches[0]), model, opt, global_step=100 + rep)
    print("--- step timeline (ms)")
    for n, t in RT.marks_report():
        print("  %7.3f  %s" % (t, n))
