import sys; sys.path.insert(0, '.')
import torch
from tests.test_bn_fold_gpu import _step
dev = torch.device('cuda:0')
for H, W, B in [(64, 128, 2), (96, 192, 3), (192, 640, 4)]:
    l0, g0, b0, k0 = _step(dev, False, H, W, B)
    l0b, g0b, _, _ = _step(dev, False, H, W, B)
    l1, g1, b1, k1 = _step(dev, True, H, W, B)
    def dev_(ga, gb):
        worst, mincos, wk = 0, 1, None
        for k in ga:
            a, b = ga[k], gb[k]
            den = a.norm().item()
            if den < 1e-12: continue
            rel = (a - b).norm().item() / den
            cos = torch.dot(a.flatten(), b.flatten()).item() / (den * b.norm().item() + 1e-30)
            if rel > worst: worst, wk = rel, k
            mincos = min(mincos, cos)
        return worst, mincos, wk
    print(H, W, B, "loss", l0, l0b, l1, "nofold-vs-nofold", dev_(g0, g0b), "fold-vs-nofold", dev_(g0, g1))
    rows = []
    for k in g0:
        a, b = g0[k], g1[k]
        den = a.norm().item()
        if den < 1e-12: continue
        rows.append(((a - b).norm().item() / den, den, k))
    rows.sort(reverse=True)
    for r in rows[:10]: print("   %.4f  |g|=%.3e  %s" % r)
