"""Is the bf16 step's gradient error zero-mean?  Average the bf16 gradients of K copies of the model whose weights carry
independent relative perturbations of 2e-4 (different rounding / ReLU decisions, same true gradient to 1e-3) and compare the
average with the fp32 oracle: an unbiased, rounding-driven error averages out (cosine rises ~ with sqrt(K) of the noise), a
systematic one does not.   python tools/probes/bf16_noise_avg.py depth H W B K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fsnet_oracle as O
from fsnet_amd.configs import meta_arch_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.utils.builder import build

depth, H, W, B, K = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
RT.tie_noise = False
sd0 = O.init_state(seed=11, depth=depth, with_pose=False)
data = O.synthetic_batch(B, H, W, seed=13)
tr = O.OracleTrainer(sd0, depth=depth, with_pose=False, clip=None)
total, ld, _, raw, _ = tr.step(data)
ddev = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}


def grads(dtype, seed):
    RT.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd = {k: (v * (1 + 2e-4 * torch.randn(v.shape, generator=g)) if (seed and v.dim() == 4) else v.clone()) for k, v in sd0.items()}
    m = build(**meta_arch_cfg(H, W, with_pose=False, depth=depth))
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    out = m(dict(ddev), dict(is_training=True))
    out["loss"].backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}


def report(tag, gs):
    gmax = max(float(r.norm()) for r in raw.values())
    enc, dec = [], []
    for k, ref in raw.items():
        if float(ref.norm()) < 1e-3 * gmax or ref.dim() != 4:
            continue
        g = gs[k]
        cos = float((g * ref).sum() / (g.norm() * ref.norm()))
        (enc if "backbone" in k else dec).append(cos)
    print("%-34s encoder mean cos %.3f (min %.3f)   decoder mean %.3f" % (tag, sum(enc) / len(enc), min(enc), sum(dec) / len(dec)))


report("fp32, perturbed weights (1 copy)", grads("fp32", 101))
acc = None
for i in range(K):
    g = grads("bf16", 200 + i)
    acc = g if acc is None else {k: acc[k] + g[k] for k in acc}
    if i + 1 in (1, 2, 4, 8, 16, 32):
        report("bf16, mean of %d perturbed copies" % (i + 1), {k: v / (i + 1) for k, v in acc.items()})
