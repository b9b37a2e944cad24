"""Calibration: what does the vendor library (MIOpen through torch.nn.functional.conv2d, bf16, channels_last,
benchmark mode) take for the encoder's 3x3/s1 shapes?  Not used by the product; a yardstick for conv3x3_halo."""
import torch, time
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
shapes = [(12, 64, 48, 160), (24, 64, 48, 160), (12, 128, 24, 80), (12, 256, 12, 40), (12, 512, 6, 20), (12, 32, 96, 320)]
for (B, C, H, W) in shapes:
    x = torch.randn(B, C, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True); w.requires_grad_(True)
    for _ in range(5):
        y = F.conv2d(x, w, padding=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        y = F.conv2d(x, w, padding=1)
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / n * 1e3
    g = torch.randn_like(y)
    for _ in range(3):
        gx, gw = torch.autograd.grad(y, (x, w), g, retain_graph=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        gx, gw = torch.autograd.grad(y, (x, w), g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * B * H * W * C * C * 9
    print("B%d C%d %dx%d: fwd %.1f us (%.0f TF/s) ; dgrad+wgrad %.1f us (%.0f TF/s)" % (B, C, H, W, tf, fl / tf / 1e6, tb, 2 * fl / tb / 1e6))
