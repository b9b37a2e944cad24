"""which parameter gradients differ between a replayed step with late issue and one without?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from oracle import fsnet_oracle as O
from tests.test_fisheye_gpu import _fisheye_model, _fisheye_batch, to_dev
from fsnet_amd.configs import training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build
dev = torch.device("cuda", 0)
B, H, W = 2, 64, 64
sd0 = O.init_state(seed=5, with_pose=False, num_out=64, max_depth=150.0)
def run(gw, late):
    RT.wgrad_late = late
    m2 = _fisheye_model(H, W, dev, sd0)
    tc = training_cfg()
    opt = build_optimizer(m2, name="adam", lr=1e-4, weight_decay=1e-5)
    hook = build(**dict(tc.training_hook, graph_warmup=gw))
    for it in range(3):
        d = _fisheye_batch(B, H, W, seed=70 + it)
        o = hook(to_dev(d, dev), m2, opt)
        torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in m2.named_parameters() if p.grad is not None}, hook.graph_replays
ge, _ = run(99, True)
gl, r1 = run(2, True)
go, r2 = run(2, False)
print("replays", r1, r2)
for k in ge:
    n = float(ge[k].norm()) + 1e-30
    a, b = float((gl[k] - ge[k]).norm()) / n, float((go[k] - ge[k]).norm()) / n
    if a > 1e-4 or b > 1e-4:
        print("%-60s late %.3e old %.3e  |g| %.3e" % (k, a, b, n))
