import sys; sys.path.insert(0, '.')
import torch
from oracle import fsnet_oracle as O
from tests.test_model_gpu import build_model, to_dev
dev = torch.device('cuda:0')
B,H,W = 4,96,320
sd0 = O.init_state(seed=3, with_pose=True)
data = O.synthetic_batch(B, H, W, seed=7)
tr = O.OracleTrainer(sd0, with_pose=True, clip=None)
total, ld, _, raw, _ = tr.step(data)
res = {}
for dt in (torch.float32, torch.bfloat16):
    m2 = build_model(True, H, W, dev, dt, sd0)
    out = m2(to_dev(data, dev), dict(is_training=True)); out["loss"].backward(); torch.cuda.synchronize()
    res[dt] = {k: p.grad.cpu().clone() for k, p in m2.named_parameters()}
names = list(res[torch.float32].keys())
sel = [n for n in names if n.endswith('conv1.weight') or n.endswith('conv2.weight') or 'sequence.0.weight' in n or n.endswith('net.0.weight') or n.endswith('net.3.weight') or 'decoder.1' in n and n.endswith('weight')]
for n in sel[:60]:
    a, b, r = res[torch.bfloat16][n], res[torch.float32][n], raw[n]
    cos = float((a*b).sum()/(a.norm()*b.norm()+1e-30))
    print("%-55s |g| %.3e  bf16-vs-fp32 relL2 %.3f cos %.4f   fp32-vs-oracle %.4f" % (n, float(b.norm()), float((a-b).norm()/b.norm()), cos, float((b-r).norm()/r.norm())))
