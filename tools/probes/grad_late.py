"""per-parameter gradient of the captured step's first replay against the eager step, for the hand-over variants"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine import nets
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build
from tests.helpers_scene import corridor_batch
dev = torch.device("cuda", 0)
RT.set_compute_dtype(torch.float32)
B, H, W = 2, 64, 128
cfg = meta_arch_cfg(H, W, with_pose=True, depth=18)
sd0 = {k: v.clone() for k, v in build(**cfg).state_dict().items()}
def run(use_graph, late=True, balance=1):
    RT.wgrad_late, RT.wgrad_balance = late, balance
    m = build(**cfg); m.load_state_dict(sd0, strict=True); m = m.to(dev).train()
    tc = training_cfg(); opt = build_optimizer(m, **tc.optimizer)
    hook = build(use_graph=use_graph, graph_warmup=2, **tc.training_hook)
    for it in range(3):
        out = hook(dict(corridor_batch(B, H, W, seed=90 + it, device=dev)[0]), m, opt)
    torch.cuda.synchronize()
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}, float(out["loss"].detach())
ge, le = run(False)
for name, kw in (("eager again", None), ("graph late+balance", dict(late=True, balance=1)), ("graph late", dict(late=True, balance=0)),
                 ("graph old", dict(late=False, balance=0))):
    g, l = run(False) if kw is None else run(True, **kw)
    gmax = max(float(v.norm()) for v in ge.values())
    rows = sorted(((float((g[k] - ge[k]).norm() / ge[k].norm()), k) for k in ge if float(ge[k].norm()) > 1e-6 * gmax), reverse=True)
    print("%-20s loss %.7f (eager %.7f); worst: %s" % (name, l, le, ", ".join("%s %.1e" % (k.replace("depth_backbone", "D").replace("pose_backbone", "P"), r) for r, k in rows[:6])))
