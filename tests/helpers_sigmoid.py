"""Seeded inputs of the sigmoid-disparity DepthDecoder case (the same construction as tools/gen_golden.py's
sigmoid_decoder_case, which made tests/golden/sigmoid_decoder.npz from the reference's class)."""
import torch

from oracle import fsnet_oracle as O


def thin(t, limit=4096):
    a = t.detach().reshape(-1)
    k = max(1, -(-a.numel() // limit))
    return a[::k].to(torch.float32).cpu().numpy().copy()


def case():
    g = torch.Generator().manual_seed(77)
    B, H, W = 2, 64, 128
    chans = [64, 64, 128, 256, 512]
    feats = [torch.randn(B, c, H >> (k + 1), W >> (k + 1), generator=g) * 0.5 for k, c in enumerate(chans)]
    sd = {k[len("head.depth_decoder."):]: v for k, v in O.init_state(seed=9, with_pose=False, num_out=1).items()
          if k.startswith("head.depth_decoder.")}
    P2 = torch.tensor([[[700.0, 0, 64, 0], [0, 700, 32, 0], [0, 0, 1, 0]], [[540.0, 0, 60, 0], [0, 540, 30, 0], [0, 0, 1, 0]]])
    wd = [torch.randn(B, 1, H >> s, W >> s, generator=g) for s in range(4)]
    wq = [torch.randn(B, 1, H >> s, W >> s, generator=g) for s in range(4)]
    return feats, sd, P2, wd, wq


def oracle_run(base_fx, dtype=torch.float64):
    feats, sd, P2, wd, wq = case()
    fl = [f.to(dtype).requires_grad_(True) for f in feats]
    sdd = {("d." + k): (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sdd.items() if v.is_floating_point() and O.is_param(k)}
    o = O.depth_decoder_forward(sdd, "d.", fl, 0.5, 100.0, P2=(P2.to(dtype) if base_fx is not None else None),
                                base_fx=base_fx, sigmoid=True)
    loss = sum((o[("depth", s, s)] * wd[s].to(dtype)).sum() * 1e-2 + (o[("disp", s)] * wq[s].to(dtype)).sum() for s in range(4))
    loss.backward()
    return o, fl, params, loss
