"""Per-shape timing of the conv kernels at the bench geometry (B=12, 192x640, R18 depth+pose)."""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SHAPES = [  # name, Ci, Co, k, stride, pad, H, W (input)
    ("stem3", 3, 64, 7, 2, 3, 192, 640), ("l1 64-64", 64, 64, 3, 1, 1, 48, 160), ("l2.0 64-128s2", 64, 128, 3, 2, 1, 48, 160),
    ("l2 128-128", 128, 128, 3, 1, 1, 24, 80), ("l3 256-256", 256, 256, 3, 1, 1, 12, 40), ("l4 512-512", 512, 512, 3, 1, 1, 6, 20),
    ("l3.0 128-256s2", 128, 256, 3, 2, 1, 24, 80), ("l4.0 256-512s2", 256, 512, 3, 2, 1, 12, 40),
    ("ds 64-128 1x1s2", 64, 128, 1, 2, 0, 48, 160), ("ds 128-256 1x1s2", 128, 256, 1, 2, 0, 24, 80), ("ds 256-512 1x1s2", 256, 512, 1, 2, 0, 12, 40), ("dec 512-256", 512, 256, 3, 1, 1, 6, 20), ("dec 512-256@12", 512, 256, 3, 1, 0, 14, 42),
    ("dec 256-128@24", 256, 128, 3, 1, 0, 26, 82), ("dec 128-64@48", 128, 64, 3, 1, 0, 50, 162), ("dec 96-32@96", 96, 32, 3, 1, 0, 98, 322),
    ("dec 32-16@96", 32, 16, 3, 1, 1, 96, 320), ("dec 16-16@192", 16, 16, 3, 1, 0, 194, 642), ("pose 256-256", 256, 256, 3, 1, 1, 6, 20),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
print("%-18s %8s | %9s %7s | %9s %7s | %9s %7s" % ("shape", "GFLOP", "fwd us", "TF", "dgrad us", "TF", "wgrad us", "TF"))
for name, Ci, Co, k, st, pad, H, W in SHAPES:
    op = ConvOp(Ci, Co, k, k, st, pad, dt, dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    op.pack(w)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    Ho, Wo = op.out_hw(H, W)
    stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
    y = op.forward(x, stats=stats)
    gy = torch.randn_like(y)
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    fl = 2.0 * B * Ho * Wo * Co * Ci * k * k
    tf = timeit(lambda: op.forward(x, out=y, stats=stats))
    td = timeit(lambda: op.dgrad(gy, H, W))
    tw = timeit(lambda: op.wgrad(gy, x, dw))
    print("%-18s %8.2f | %9.1f %7.1f | %9.1f %7.1f | %9.1f %7.1f" % (name, fl / 1e9, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12))
