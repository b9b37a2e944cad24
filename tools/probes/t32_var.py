"""one conv shape, one kernel variant (see t32_ab.py), forward + dgrad launches, plus a same-size copy as a yardstick.
args: Ci Co H W B variant (old | 1 | 2 | 3)"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
v = sys.argv[6]
if v == "old":
    os.environ["FSNET_AMD_T32"] = "0"
else:
    os.environ["FSNET_AMD_T32"] = "1"; os.environ["FSNET_AMD_T32_CFG"] = v
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
for _ in range(12):
    op.forward(x, out=y, stats=stats)
torch.cuda.synchronize()
for _ in range(6):
    y.copy_(x)
torch.cuda.synchronize()
