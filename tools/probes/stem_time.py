"""alone-times of the stem's fused passes at the benchmark size (36 images as 12 + 24, 96x320x64, bf16)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from fsnet_amd.hip import ops
from fsnet_amd.hip.conv import run_specs
from fsnet_amd.engine.nets import bn_tensors
dev = torch.device("cuda")
dt = torch.bfloat16
C, H, W = 64, 96, 320
def mk(N, G):
    x = torch.randn(N, H, W, C, device=dev).to(dt)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    stats = torch.zeros(G, 8, 2, C, dtype=torch.float64, device=dev)
    n = N // G
    for g in range(G):
        v = x[g * n:(g + 1) * n].double()
        stats[g, 0, 0] = v.sum((0, 1, 2)); stats[g, 0, 1] = (v * v).sum((0, 1, 2))
    return dict(x=x, bn=bn, stats=stats if G > 1 else stats[0], G=G, n=n, N=N)
L = [mk(12, 1), mk(24, 2)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
sts = [ops.BnState(C, dev, l["G"]) for l in L]
ys = [torch.empty_like(L[0]["x"]), None]
pools = [(torch.empty(l["N"], H // 2, W // 2, C, device=dev, dtype=dt), torch.empty(l["N"], H // 2, W // 2, C, device=dev, dtype=torch.uint8)) for l in L]
def fwd_fused():
    run_specs([ops.bn_apply_spec(l["x"], l["stats"], bn_tensors(l["bn"]), st, y, H, W, l["n"] * H * W, relu=True, groups=l["G"], pool=pl)
               for l, st, y, pl in zip(L, sts, ys, pools)])
ys2 = [torch.empty_like(l["x"]) for l in L]
def fwd_sep():
    run_specs([ops.bn_apply_spec(l["x"], l["stats"], bn_tensors(l["bn"]), st, y, H, W, l["n"] * H * W, relu=True, groups=l["G"])
               for l, st, y in zip(L, sts, ys2)])
    ops.maxpool_fwd_multi(ys2)
print("forward  fused %.1f us   separate (bn_apply + maxpool) %.1f us" % (timeit(fwd_fused), timeit(fwd_sep)))
dpool = [torch.randn_like(p[0]) for p in pools]
add = [torch.randn_like(L[0]["x"]), None]
dxs = [torch.empty_like(l["x"]) for l in L]
dg = [torch.zeros(C, device=dev) for _ in L]; db = [torch.zeros(C, device=dev) for _ in L]
def calls(pool):
    out = []
    for k, l in enumerate(L):
        d = dict(x=l["x"], gamma=l["bn"].weight.data, st=sts[k], dx=dxs[k], dgamma=dg[k], dbeta=db[k], H=H, W=W, relu=True)
        if pool:
            d.update(dout=add[k], y=None, pool=(dpool[k], pools[k][1], l["bn"].bias.data))
        else:
            d.update(dout=d0[k], y=ys2[k])
        out.append(d)
    return out
def bwd_fused(phase):
    ops.bn_backward_multi(calls(True), phase=phase)
d0 = ops.maxpool_bwd_multi(dpool, [p[1] for p in pools], H, W, add)
def bwd_sep_pool():
    ops.maxpool_bwd_multi(dpool, [p[1] for p in pools], H, W, add)
def bwd_sep(phase):
    ops.bn_backward_multi(calls(False), phase=phase)
print("backward fused: reduce %.1f us, reduce+apply %.1f us" % (timeit(lambda: bwd_fused("reduce")), timeit(lambda: bwd_fused("all"))))
print("backward separate: maxpool_bwd %.1f us, reduce %.1f us, reduce+apply %.1f us" % (timeit(bwd_sep_pool), timeit(lambda: bwd_sep("reduce")), timeit(lambda: bwd_sep("all"))))
