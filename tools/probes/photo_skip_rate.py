"""How often could photo_fused_bwd_kernel skip a row step for a frame?  A (scale, frame) pass contributes to output row q of a
60-column strip only where a pixel of rows q-1..q+1 x columns x-1..x+1 selected that frame (sel == 2 + f).  Measured from the
forward's selection bytes: on the bench's synthetic triplets (untrained weights) and on the corridor scene before / after
training with dataset poses.     python tools/probes/photo_skip_rate.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.nn.functional as F
import bench
from fsnet_amd.configs import meta_arch_cfg, training_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
from fsnet_amd.vision_base.utils.builder import build
from tests.helpers_scene import corridor_batch

dev = torch.device("cuda", 0)


def rates(sel, BW=60):
    """sel: [S,B,H,W] u8 -> per frame: fraction of pixels selecting it, fraction of (strip, row) steps with no selecting pixel
    in the 3-row x (BW+2)-column neighbourhood, and the same for whole 32-row strips"""
    S, B, H, W = sel.shape
    out = []
    for f in (0, 1):
        m = (sel == 2 + f).float().view(S * B, 1, H, W)
        pix = float(m.mean())
        rows3 = F.max_pool2d(m, (3, 1), stride=1, padding=(1, 0))                  # any in rows q-1..q+1
        nx = (W + BW - 1) // BW
        pad = nx * BW - W
        r = F.pad(rows3, (1, 1 + pad, 0, 0))
        strips = torch.stack([r[..., i * BW:i * BW + BW + 2].amax(-1) for i in range(nx)], -1)   # [SB,1,H,nx]
        row_skip = 1.0 - float(strips.mean())
        ny = (H + 31) // 32
        st = F.pad(strips, (0, 0, 0, ny * 32 - H))
        whole = torch.stack([st[:, :, j * 32:(j + 1) * 32].amax(2) for j in range(ny)], 2)
        out.append((pix, row_skip, 1.0 - float(whole.mean())))
    return out


def report(name, model):
    sel = model.head._pl.sel
    ident = float((sel < 2).float().mean())
    r = rates(sel)
    print("%-46s identity wins %.3f | frame +1: pixels %.3f, skippable row steps %.3f, whole strips %.3f | frame -1: %.3f, %.3f, %.3f"
          % (name, ident, *r[0], *r[1]), flush=True)


RT.set_compute_dtype("bf16")
model = build(**meta_arch_cfg(192, 640, with_pose=True)).to(dev).train()
tc = training_cfg()
opt = build_optimizer(model, **tc.optimizer)
hook = build(use_graph=False, **tc.training_hook)
batches = bench.synthetic_device_batches(12, 192, 640, dev, 0)
hook(dict(batches[0]), model, opt)
report("bench triplets, untrained R18 depth+pose", model)
for i in range(60):
    hook(dict(batches[i % 4]), model, opt)
report("bench triplets, after 60 steps", model)

H, W, B = 192, 640, 4
m = build(**meta_arch_cfg(H, W, with_pose=False)).to(dev).train()
opt = build_optimizer(m, **tc.optimizer)
hook = build(**tc.training_hook)
pool = [corridor_batch(B, H, W, seed=4000 + i, device=dev)[0] for i in range(32)]
hook(dict(pool[0]), m, opt)
torch.cuda.synchronize()
report("corridor scene 192x640, untrained R18, dataset poses", m)
for i in range(600):
    hook(dict(pool[i % 32]), m, opt)
torch.cuda.synchronize()
report("corridor scene, after 600 steps", m)
