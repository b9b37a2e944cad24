"""weight gradient of the encoder's 3x3 layers alone (graph replay of 20 launches): us per launch.  Run once per
FSNET_AMD_WG_PF2 setting (the switch is read once per process)."""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
dev = torch.device('cuda:0'); dt = torch.bfloat16
SHAPES = [(64, 64, 48, 160, 12), (64, 64, 48, 160, 24), (128, 128, 24, 80, 12), (256, 256, 12, 40, 12), (512, 512, 6, 20, 12),
          (64, 64, 80, 256, 8)]
for Ci, Co, H, W, B in SHAPES:
    op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
    x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
    gy = torch.randn(B, H, W, op.Co_p, device=dev).to(dt)
    dw = torch.zeros(Co, Ci, 3, 3, device=dev)
    for _ in range(3): op.wgrad(gy, x, dw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): op.wgrad(gy, x, dw)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"PF2={os.environ.get('FSNET_AMD_WG_PF2','1')}  {Ci}->{Co} @{H}x{W} B={B}: {e0.elapsed_time(e1) * 1000 / 200:.1f} us (kernel + reduce)")
