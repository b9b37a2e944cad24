"""One captured graph per chain: does the backward's pose branch start at once when the two chains' backward passes are
separate graphs launched on two real streams (docs/LAB_r06.md: inside ONE graph the executor releases it 0.9 ms late)?
    python tools/probes/two_graphs.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.nets import join_companions_final
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build

dev = torch.device("cuda", 0)
RT.set_compute_dtype("bf16")
model = build(**meta_arch_cfg(192, 640, with_pose=True)).to(dev).train()
tc = training_cfg(clip_gradients=35.0, lr=1e-4)
opt = build_optimizer(model, **tc.optimizer)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
K = 200

# ---- reference: the shipped single graph
hook = build(**tc.training_hook)
for i in range(12):
    hook(dict(batches[i % 4]), model, opt, global_step=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    hook(dict(batches[i % 4]), model, opt, global_step=12 + i)
torch.cuda.synchronize()
print("one graph:            %.3f ms/step" % ((time.perf_counter() - t0) / K * 1e3), flush=True)

# ---- four graphs: forward + loss (+ its backward) | depth backward | pose backward | clip + Adam
main = RT.new_stream(dev)
side = RT.side_stream(dev)
arena = model.ensure_arena()
meta = dict(epoch_num=0, global_step=0, is_training=True)
static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batches[0].items()}
eager = build(use_graph=False, **tc.training_hook)
cur = torch.cuda.current_stream()
main.wait_stream(cur)
with torch.cuda.stream(main):
    for i in range(3):
        eager(dict(static), model, opt)
    opt.sync_lr()
torch.cuda.synchronize()
gA, gB, gC, gD = (torch.cuda.CUDAGraph() for _ in range(4))
with torch.cuda.graph(gA, stream=main):
    arena.zero_grads(lazy=True)
    out = model(dict(static), meta)
    loss = out["loss"]
    Ts, dd = model.head._loss_inputs
    grads = torch.autograd.grad(loss, list(Ts) + list(dd), allow_unused=True)
gT, gdd = grads[:2], grads[2:]
keep = [(t, g) for t, g in zip(dd, gdd) if g is not None]
with torch.cuda.graph(gB, stream=main):
    torch.autograd.backward([t for t, _ in keep], [g for _, g in keep])
    join_companions_final()
with torch.cuda.graph(gC, stream=side):
    torch.autograd.backward(list(Ts), list(gT))
    join_companions_final()
steps_before = opt._step_count_fused
with torch.cuda.graph(gD, stream=main):
    opt.step(max_norm=35.0, grad_scale=1.0)
torch.cuda.synchronize()
evA, evC = torch.cuda.Event(), torch.cuda.Event()


def step(i):
    with torch.cuda.stream(main):
        for k, v in batches[i % 4].items():
            if isinstance(v, torch.Tensor):
                static[k].copy_(v, non_blocking=True)
        opt.prepare_replay()
        gA.replay()
        evA.record(main)
    side.wait_event(evA)
    with torch.cuda.stream(side):
        gC.replay()
        evC.record(side)
    with torch.cuda.stream(main):
        gB.replay()
        main.wait_event(evC)
        gD.replay()
        opt.note_step()
        RT.bump_weights()


for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
print("four graphs, 2 streams: %.3f ms/step   (loss %.5f)" % ((time.perf_counter() - t0) / K * 1e3, float(model.head._pl.out[-1])), flush=True)
