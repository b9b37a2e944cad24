"""One conv shape, a few launches of forward / dgrad (for rocprofv3 passes).  args: Ci Co H W B [k]"""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, k, k, 1, k // 2, dt, dev)
op.pack(torch.randn(Co, Ci, k, k, device=dev) * 0.05)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
import os
stats = None if os.environ.get('NOSTATS') else torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = op.forward(x, stats=stats)
gy = torch.randn_like(y)
for _ in range(6):
    op.forward(x, out=y, stats=stats)
for _ in range(6):
    op.dgrad(gy, H, W)
torch.cuda.synchronize()
