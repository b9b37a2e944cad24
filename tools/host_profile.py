"""cProfile of the host side of eager training steps (where do the ~8 ms of Python per step go?)"""
import cProfile, pstats, os, sys, io
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
if os.environ.get("FSNET_PROF_DP", "0") != "0":     # data-parallel bookkeeping at world size 1 over RCCL
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from fsnet_amd.engine.dataparallel import DataParallelContext
    model.ensure_arena()
    RT.dp = DataParallelContext(model)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
for i in range(4):
    hook(dict(batches[i % len(batches)]), model, opt)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 20
torch.autograd.set_multithreading_enabled(False)   # backward on this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    hook(dict(batches[i % len(batches)]), model, opt)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time per step (issue only, profiler on): %.2f ms" % ((t1 - t0) / N * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
