"""One 1x1 conv shape, a few launches of forward (no statistics) / forward with statistics / dgrad (for rocprofv3
passes).  args: Ci Co H W B [stride]"""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
st = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, 1, 1, st, 0, dt, dev)
op.pack(torch.randn(Co, Ci, 1, 1, device=dev) * 0.05)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = op.forward(x)
gy = torch.randn_like(y)
for _ in range(6):
    op.forward(x, out=y)
torch.cuda.synchronize()
for _ in range(6):
    op.forward(x, out=y, stats=stats)
torch.cuda.synchronize()
if st == 1:
    for _ in range(6):
        op.dgrad(gy, H, W)
torch.cuda.synchronize()
