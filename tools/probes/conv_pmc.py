"""Runs a handful of conv forward launches (for rocprofv3 --pmc): l1, l2, l3, l4 shapes, bf16, B=12."""
import sys; sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
for name, Ci, Co, H, W in (("l1", 64, 64, 48, 160), ("l2", 128, 128, 24, 80), ("l3", 256, 256, 12, 40), ("l4", 512, 512, 6, 20)):
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
    x = torch.randn(12, H, W, op.Ci_p, device=dev).to(dt)
    y = op.forward(x)
    for _ in range(5):
        op.forward(x, out=y)
    gy = torch.randn_like(y)
    dw = torch.zeros(Co, Ci, 3, 3, device=dev)
    for _ in range(3):
        op.wgrad(gy, x, dw)
torch.cuda.synchronize()
