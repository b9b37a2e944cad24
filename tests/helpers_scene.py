"""A geometrically consistent synthetic scene for training tests: three frames of ONE rigid world
rendered exactly by ray casting, so that the photometric minimum of the reference's loss
(monodepth2_decoder.py:61-116, 205-304: BackprojectDepth -> Project3D -> grid_sample -> SSIM + L1, per-pixel minimum) exists
and sits at the generating depth and camera motion — unlike the throughput batches of SURVEY 8(d), whose frames are
shifted noise-like fields no depth can explain.

World = a closed corridor in the target camera's frame (x right, y down, z forward; KITTI-like pinhole intrinsics of SURVEY
8(d)): ground plane y = h, side walls x = -a and x = +b, end wall z = zf, each carrying a band-limited sinusoid texture in
its own surface coordinates.  Frame f is rendered from the camera at c_f with yaw, pixel by pixel: ray -> nearest plane ->
texture, each texture component attenuated by the pixel's footprint on the surface (no aliasing that differs between the
frames).  Returned with the batch (SURVEY 8b contract): ('relative_pose', f) = T(cam_0 -> cam_f) as the dataset would store
it, and the target frame's exact depth map.

Test infrastructure only (tests/): plain torch, any device."""
import math

import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _texture(s, t, foot, params):
    """[B,H,W] surface coordinates (metres) -> [B,3,H,W]: sum of K plane waves per channel, amplitude rolled off where the
    wavelength approaches the pixel footprint `foot` (metres on the surface)"""
    omega, theta, phase, amp = params            # [B,K], [B,K], [B,3,K], [B,3,K]
    u = s.unsqueeze(-1) * torch.cos(theta)[:, None, None, :] + t.unsqueeze(-1) * torch.sin(theta)[:, None, None, :]
    arg = u * omega[:, None, None, :]            # [B,H,W,K]
    att = torch.exp(-0.5 * (0.6 * foot.unsqueeze(-1) * omega[:, None, None, :]) ** 2)
    col = [(amp[:, c, None, None, :] * att * torch.sin(arg + phase[:, c, None, None, :])).sum(-1) for c in range(3)]
    return torch.stack(col, 1)


def _tex_params(B, gen, device, scale, K=10):
    lam = scale * torch.exp(torch.empty(B, K).uniform_(math.log(0.35), math.log(7.0), generator=gen))
    omega = 2 * math.pi / lam
    theta = torch.empty(B, K).uniform_(0, math.pi, generator=gen)
    phase = torch.empty(B, 3, K).uniform_(0, 2 * math.pi, generator=gen)
    amp = torch.empty(B, 3, K).uniform_(0.4, 1.0, generator=gen)
    amp = 0.42 * amp / amp.pow(2).sum(-1, keepdim=True).sqrt()
    return tuple(x.to(device) for x in (omega, theta, phase, amp))


def corridor_batch(B, H, W, seed=0, device="cpu", frame_ids=(0, 1, -1), fixed_geometry=False, scale=1.0):
    """-> (batch dict per SURVEY 8(b), truth dict: 'depth' [B,1,H,W] of the target frame, ('T', f) [B,4,4]).
    scale: the unit of length (1.0: metres, a KITTI-sized corridor driven through at 0.5-1.1 m per frame).  The images do
    not depend on it; depth and translation do — a pose network that starts from random weights puts out translations of
    ~0.01 units (pose_decoder.py:45: 0.01 * out), and with the world in metres would have to grow them a hundredfold."""
    gen = torch.Generator().manual_seed(int(seed))
    dev = torch.device(device)

    def U(lo, hi):
        return torch.empty(B).uniform_(lo, hi, generator=gen).to(dev)
    if fixed_geometry:
        h, a, b, zf = (torch.full((B,), v * scale, device=dev) for v in (1.65, 4.5, 5.5, 45.0))
    else:
        h, a, b, zf = U(1.4, 1.9) * scale, U(3.0, 7.0) * scale, U(3.0, 7.0) * scale, U(30.0, 60.0) * scale
    fx, fy, cx, cy = 0.58 * W, 1.92 * H, 0.5 * W, 0.5 * H
    tex = [_tex_params(B, gen, dev, scale) for _ in range(4)]                    # ground, left, right, end wall
    base = torch.empty(B, 3, 4).uniform_(0.35, 0.65, generator=gen).to(dev)   # mean colour per surface
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev),
                            indexing="ij")
    dcx, dcy = ((xs - cx) / fx).expand(B, H, W), ((ys - cy) / fy).expand(B, H, W)   # camera-frame ray, z component 1
    batch, truth = {}, {}
    mean = torch.tensor(MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(STD, device=dev).view(1, 3, 1, 1)
    step, lat, yaw = U(0.5, 1.1) * scale, U(-0.05, 0.05) * scale, U(-0.015, 0.015)
    for f in frame_ids:
        # camera f: moved by f * (lat, 0, step) and turned by f * yaw about the vertical axis
        c = torch.stack([f * lat, torch.zeros_like(lat), f * step], 1)               # [B,3] position in the world
        ang = f * yaw
        cs, sn = torch.cos(ang)[:, None, None], torch.sin(ang)[:, None, None]
        # world direction = R_f d_c, R_f = rotation about y by ang
        dx = cs * dcx + sn
        dy = dcy
        dz = -sn * dcx + cs
        big = torch.full_like(dx, 1e9)
        cxw, cyw, czw = (c[:, i, None, None] for i in range(3))
        t_g = torch.where(dy > 1e-6, (h[:, None, None] - cyw) / dy.clamp_min(1e-6), big)
        t_l = torch.where(dx < -1e-6, (-a[:, None, None] - cxw) / dx.clamp_max(-1e-6), big)
        t_r = torch.where(dx > 1e-6, (b[:, None, None] - cxw) / dx.clamp_min(1e-6), big)
        t_e = torch.where(dz > 1e-6, (zf[:, None, None] - czw) / dz.clamp_min(1e-6), big)
        ts = torch.stack([t_g, t_l, t_r, t_e], 1)                                    # [B,4,H,W]
        t, which = ts.min(1)
        px, py, pz = cxw + t * dx, cyw + t * dy, czw + t * dz
        norm = torch.sqrt(dx * dx + dy * dy + dz * dz)
        # pixel footprint on the surface: range / focal length, stretched by the grazing angle
        inc = torch.stack([dy.abs(), dx.abs(), dx.abs(), dz.abs()], 1).gather(1, which.unsqueeze(1)).squeeze(1) / norm
        foot = t * norm / fx / inc.clamp_min(0.05)
        surf = [(px, pz), (py, pz), (py, pz), (px, py)]
        img = torch.zeros(B, 3, H, W, device=dev)
        for k in range(4):
            m = (which == k).unsqueeze(1).float()
            img = img + m * (base[:, :, k, None, None] + _texture(surf[k][0], surf[k][1], foot, tex[k]))
        img = img.clamp(0, 1)
        batch[("original_image", f)] = img
        batch[("image", f)] = (img - mean) / std
        if f == 0:
            truth["depth"] = t.unsqueeze(1)              # camera-frame z: the ray's z component is 1 and R_0 = I
        else:
            # T(cam_0 -> cam_f): X_f = R_f^T (X - c_f)
            T = torch.zeros(B, 4, 4, device=dev)
            cs1, sn1 = torch.cos(ang), torch.sin(ang)
            Rt = torch.zeros(B, 3, 3, device=dev)
            Rt[:, 0, 0], Rt[:, 0, 2], Rt[:, 1, 1], Rt[:, 2, 0], Rt[:, 2, 2] = cs1, -sn1, 1.0, sn1, cs1
            T[:, :3, :3] = Rt
            T[:, :3, 3] = -(Rt @ c.unsqueeze(-1)).squeeze(-1)
            T[:, 3, 3] = 1.0
            truth[("T", f)] = T
            batch[("relative_pose", f)] = T.clone()
    P2 = torch.zeros(B, 3, 4, device=dev)
    P2[:, 0, 0], P2[:, 0, 2], P2[:, 1, 1], P2[:, 1, 2], P2[:, 2, 2] = fx, cx, fy, cy, 1.0
    batch["P2"] = P2
    batch["patched_mask"] = torch.ones(B, H, W, dtype=torch.float64, device=dev)
    return batch, truth


def log_depth_correlation(pred_depth, true_depth):
    """Pearson correlation of log depth per image, averaged (scale-free: a learned pose fixes depth only up to scale)"""
    a = torch.log(pred_depth.flatten(1).double().clamp_min(1e-3))
    b = torch.log(true_depth.flatten(1).double().clamp_min(1e-3))
    a = a - a.mean(1, keepdim=True)
    b = b - b.mean(1, keepdim=True)
    return float(((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1) + 1e-12)).mean())
