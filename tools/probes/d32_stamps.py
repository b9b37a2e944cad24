"""cycle stamps of block 0 of the dual-group conv kernel (development).  args: Ci Co H W B cfg [abl]"""
import os, sys
sys.path.insert(0, '.')
import torch
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, B = (int(v) for v in sys.argv[1:6])
os.environ["FSNET_AMD_T32"] = "1"; os.environ["FSNET_AMD_D32"] = "1"; os.environ["FSNET_AMD_D32_CFG"] = sys.argv[6]
if len(sys.argv) > 7:
    os.environ["FSNET_AMD_T32_ABL"] = sys.argv[7]
dev = torch.device('cuda:0'); dt = torch.bfloat16
op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
op.pack(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
x = torch.randn(B, H, W, op.Ci_p, device=dev).to(dt)
stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = torch.empty(B, H, W, op.Co_p, dtype=dt, device=dev)
dbg = torch.zeros(128, dtype=torch.int64, device=dev)
for _ in range(3):
    op.forward(x, out=y, stats=stats)
torch.cuda.synchronize()
os.environ["FSNET_AMD_T32_DBG"] = str(dbg.data_ptr())
op.forward(x, out=y, stats=stats)
torch.cuda.synchronize()
d = dbg.cpu().tolist()
for g in range(2):
    v = [c for c in d[g * 64:(g + 1) * 64] if c]
    print("group", g, "stamps (cycles since first):", [c - v[0] for c in v])
