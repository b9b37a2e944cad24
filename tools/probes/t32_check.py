"""where does the 32x32-tile conv kernel differ from F.conv2d?  args: Ci Co H W N [dtype]"""
import sys
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from fsnet_amd.hip.conv import ConvOp
Ci, Co, H, W, N = (int(v) for v in sys.argv[1:6])
dt = torch.float32 if len(sys.argv) > 6 and sys.argv[6] == "f32" else torch.bfloat16
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Ci, H, W, generator=g); w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
if dt == torch.bfloat16:
    x = x.bfloat16().float(); w = w.bfloat16().float()
ref = F.conv2d(x, w, None, padding=1)
op = ConvOp(Ci, Co, 3, 3, 1, 1, dt, dev)
op.pack(w.to(dev).contiguous())
xd = x.to(dev).permute(0, 2, 3, 1).contiguous().to(dt)
stats = torch.zeros(8, 2, op.Co_p, dtype=torch.float64, device=dev)
y = op.forward(xd, stats=stats, out_f32=True)
torch.cuda.synchronize()
err = (y.permute(0, 3, 1, 2).float().cpu() - ref).abs()
print("max err", err.max().item(), "scale", ref.abs().max().item())
print("per channel max err:", [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()])
print("per image:", err.amax(dim=(1, 2, 3)).tolist())
e2 = err.amax(dim=(0, 1))
print("rows with err:", (e2.amax(1) > 1e-2).nonzero().flatten().tolist()[:40])
print("cols with err:", (e2.amax(0) > 1e-2).nonzero().flatten().tolist()[:40])
s = stats.sum(0)
print("stat1 err", (s[0, :Co].cpu() - ref.double().sum((0, 2, 3))).abs().max().item())
