"""1x1 convolutions of ResNet-50 at 320x1024, B=8: forward / data gradient per shape."""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp, USE_1X1
dev = torch.device('cuda:0'); dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("l1 64-64", 64, 64, 1, 80, 256), ("l1 64-256", 64, 256, 1, 80, 256), ("l1 256-64", 256, 64, 1, 80, 256),
          ("l2 256-128", 256, 128, 1, 80, 256), ("l2 128-512", 128, 512, 1, 40, 128), ("l2 512-128", 512, 128, 1, 40, 128),
          ("l3 256-1024", 256, 1024, 1, 20, 64), ("l3 1024-256", 1024, 256, 1, 20, 64), ("l4 512-2048", 512, 2048, 1, 10, 32),
          ("l4 2048-512", 2048, 512, 1, 10, 32), ("ds 256-512 s2", 256, 512, 2, 80, 256), ("ds 1024-2048 s2", 1024, 2048, 2, 20, 64)]
def timeit(fn, n=20):
    """n launches replayed as one hipGraph (the launches are shorter than the host's issue time)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e-3
print("conv1x1 kernel:", USE_1X1)
for name, Ci, Co, st, H, W in SHAPES:
    op = ConvOp(Ci, Co, 1, 1, st, 0, dt, dev)
    op.pack(torch.randn(Co, Ci, 1, 1, device=dev) * 0.05)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    Ho, Wo = op.out_hw(H, W)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(x, stats=stats)
    gy = torch.randn_like(y)
    fl = 2.0 * B * Ho * Wo * Co * Ci
    mb = (B * Ho * Wo * (Ci + Co) * 2) / 1e6
    tf = timeit(lambda: op.forward(x, out=y, stats=stats))
    tn = timeit(lambda: op.forward(x, out=y))
    td = timeit(lambda: op.dgrad(gy, H, W)) if st == 1 else float('nan')
    tdf = float('nan')
    if st == 1:     # the Bottleneck's conv1 data gradient: identity gradient added, previous block's ReLU mask and BatchNorm sums
        from fsnet_amd.hip import ops
        cprev, yprev, addend = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
        bst = ops.BnState(op.Ci_p, dev, 1)
        bst.mean.normal_(0, 0.1); bst.invstd.uniform_(0.5, 1.5); bst.count = float(B * H * W)
        sums = torch.zeros(8, 2, op.Ci_p, dtype=torch.float64, device=dev)
        if op.can_fuse_bn_bwd(B, H, W, 1):
            tdf = timeit(lambda: op.dgrad(gy, H, W, addend=addend, mask=yprev, bn_fuse=(cprev, bst, sums)))
    dw = torch.zeros(Co, Ci, 1, 1, device=dev)
    tw = timeit(lambda: op.wgrad(gy, x, dw))
    print("%-16s %6.2f GF %6.1f MB | fwd %6.1f us %6.1f TF %5.2f TB/s | no stats %6.1f us | dgrad %6.1f us %6.1f TF | dgrad+add+mask+bn %6.1f us | wgrad %6.1f us %6.1f TF" % (name, fl / 1e9, mb, tf * 1e6, fl / tf / 1e12, mb / tf / 1e6, tn * 1e6, td * 1e6, fl / td / 1e12, tdf * 1e6, tw * 1e6, fl / tw / 1e12))
