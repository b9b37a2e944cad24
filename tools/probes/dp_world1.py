"""What the data-parallel bookkeeping costs per step, measured at world size 1 over RCCL: every SyncBN all-reduce and
the gradient buckets are issued (RCCL runs them as single-rank collectives), so the difference to the plain step is
host time + per-collective launch latency — a lower bound on the N > 1 step.
    python tools/probes/dp_world1.py [graph]"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.distributed as dist

import bench
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
use_dp = "nodp" not in sys.argv
graph = "graph" in sys.argv
RT.set_compute_dtype("bf16")
model = build(**meta_arch_cfg(192, 640, with_pose=True)).to(dev).train()
tc = training_cfg(clip_gradients=35.0, lr=1e-4)
opt = build_optimizer(model, **tc.optimizer)
hook = build(use_graph=graph, graph_dp=graph, **tc.training_hook)
if use_dp:
    from fsnet_amd.engine.dataparallel import DataParallelContext
    RT.dp = DataParallelContext(model)
    n = [0]
    orig = RT.dp.allreduce_small

    def counted(t, out=None):
        n[0] += 1
        orig(t, out)
    RT.dp.allreduce_small = counted
    if "fakecomm" in sys.argv:
        # RCCL runs a single-rank collective as nothing (in place) or a copy: stand-in kernels, so that the captured step has
        # the nodes a multi-rank step has (graph structure / executor stream assignment; not their duration)
        def fake(t, out=None):
            if out is None:
                t.mul_(1.0)
            else:
                torch.mul(t, 1.0, out=out)
        RT.dp._direct.all_reduce_sum = fake
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
pre = 0
while use_dp and graph and not hook.tune_done and pre < 200:       # encoder-pass autotune (FSNET_AMD_LANES=auto)
    hook(dict(batches[pre % len(batches)]), model, opt, global_step=pre)
    pre += 1
for i in range(8):
    hook(dict(batches[i % len(batches)]), model, opt, global_step=i)
torch.cuda.synchronize()
if use_dp:
    n[0] = 0
t0 = time.perf_counter()
K = 30
for i in range(K):
    hook(dict(batches[i % len(batches)]), model, opt, global_step=8 + i)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("dp=%s graph=%s: %.3f ms/step, %.1f samples/s, small collectives/step: %s" % (
    use_dp, graph, el / K * 1e3, 12 * K / el, (n[0] / K) if use_dp else "-"))
print("encoder pass: %s, weight gradients %s, autotune: %s after %d steps" % (
    "lanes" if RT.lanes else "chains", RT.dp.wgrad_mode if RT.dp is not None else "-", RT.encoder_pass_ms, pre))
dist.destroy_process_group()
