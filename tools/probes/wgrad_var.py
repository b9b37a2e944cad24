"""one weight-gradient shape, a few launches (kernel-trace runs).  args: Ci Co H W B"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
gy = torch.randn(B, H, W, op.Co_p, device=dev).to(dt)
dw = torch.zeros(Co, Ci, 3, 3, device=dev)
for _ in range(10):
    op.wgrad(gy, x, dw)
torch.cuda.synchronize()
