"""one conv shape on the 32x32-tile kernel, a few launches (for rocprofv3 passes).  args: Ci Co H W B cfg [mode]"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
os.environ["FSNET_AMD_T32_CFG"] = sys.argv[6] if len(sys.argv) > 6 else "3"
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
for _ in range(10):
    op.forward(x, out=y, stats=stats)
torch.cuda.synchronize()
