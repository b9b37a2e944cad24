"""CPU oracle for the fisheye (Mei unified camera model) variant of the photometric chain — BASELINE configs[3].

THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT (same rules as oracle/fsnet_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Restates, with the reference file:line each function follows:
  mei_lut                 MeiCameraProjection.image2cam cache fill (mei_fisheye_utils.py:140-166) with the
                          numba kernels newton_methods / bisection_methods / whole_map_backtracking (:66-120)
  cam2image               _cam2image + mei_distort (mei_fisheye_utils.py:14-51)
  image2cam               MeiCameraProjection.image2cam (:131-187)
  photometric_loss        FishEyeDecoder._generate_images_pred (monodepth2_decoder.py:355-411) followed by
                          MonoDepth2Decoder.compute_total_reprojection_loss / loss (:205-347)
Pinned by tests/golden/fisheye.npz (tools/gen_golden.py::gen_fisheye runs the real classes).

numba compiles the three root finders with float64 arithmetic (the float32 radius read from the array is promoted
when it meets the float64 calibration scalars); this restatement does the same in numpy float64, the element-wise
loop vectorised over pixels with an "already returned" mask.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import fsnet_oracle as O


def _radial(k1, k2, r1, r0):
    """radial_distort_func (mei_fisheye_utils.py:66-68); r0**2 / r0**4 as numba lowers them (multiplications)."""
    r2 = r0 * r0
    return r0 - r1 / (1 + k1 * r2 + k2 * (r2 * r2))


def _newton(k1, k2, x0, tol=1e-6, max_iter=100):
    """newton_methods (:70-79), vectorised: a pixel stops updating at the iteration where |f| < tol."""
    x = x0.copy()
    done = np.zeros(x.shape, dtype=bool)
    for _ in range(max_iter):
        f = _radial(k1, k2, x0, x)
        done |= np.abs(f) < tol
        if done.all():
            break
        df = (_radial(k1, k2, x0, x + tol) - f) / tol
        with np.errstate(divide="ignore", invalid="ignore"):
            xn = x - f / df
        x = np.where(done, x, xn)
    return x


def _mirror(r0, xi, Z):
    """mirror_backtrack_func (:81-83)"""
    return r0 * r0 - (1 - Z * Z) / ((xi + Z) * (xi + Z))


def _bisect(r0, xi, tol=1e-6, max_iter=100):
    """bisection_methods(r0, xi, 0, 1) (:85-101): (flag, Z); no sign change -> (False, -1)."""
    x0 = np.zeros_like(r0)
    x1 = np.ones_like(r0)
    y0, y1 = _mirror(r0, xi, x0), _mirror(r0, xi, x1)
    bad = y0 * y1 > 0
    x = np.zeros_like(r0)
    done = bad.copy()
    for _ in range(max_iter):
        xm = (x0 + x1) / 2
        f = _mirror(r0, xi, xm)
        x = np.where(done, x, xm)
        done_now = (~done) & (np.abs(f) < tol)
        done |= done_now
        if done.all():
            break
        left = f * _mirror(r0, xi, x0) < 0
        upd = ~done
        x1 = np.where(upd & left, xm, x1)
        x0 = np.where(upd & ~left, xm, x0)
    flag = ~bad
    return flag, np.where(bad, -1.0, x)


def mei_lut(H, W, gamma1, gamma2, u0, v0, k1, k2, xi):
    """(X, Y, Z, mask) float32 [H, W]: the per-calibration table of image2cam (:150-166).  X, Y, r1 are float32
    like the reference's numpy expressions; the root finders run in float64 and store float32."""
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    X = (xs.astype(np.float32) - np.float32(u0)) / np.float32(gamma1)
    Y = (ys.astype(np.float32) - np.float32(v0)) / np.float32(gamma2)
    r1 = np.sqrt(X ** 2 + Y ** 2)                       # float32
    r0 = _newton(float(k1), float(k2), r1.astype(np.float64))
    flag, Zd = _bisect(r0, float(xi))
    Z = Zd.astype(np.float32)
    mask = flag.astype(np.float32)
    mask[Z < np.float32(0.05)] = 0
    nm = np.logical_not(mask)
    Z[nm] = -1
    X[nm] = -1
    Y[nm] = -1
    X = X * (Z + np.float32(xi))
    Y = Y * (Z + np.float32(xi))
    return X, Y, Z, mask


def cam2image(points, P, calib):
    """_cam2image (:23-51) for a float32 tensor [..., 3]; P: [3, 4] tensor; returns (u, v) in pixels."""
    k1, k2 = calib["distortion_parameters"]["k1"], calib["distortion_parameters"]["k2"]
    xi = calib["mirror_parameters"]["xi"]
    eps = 1e-6
    norm = torch.norm(points, dim=-1)
    x = points[..., 0] / (norm + eps)
    y = points[..., 1] / (norm + eps)
    z = points[..., 2] / (norm + eps)
    x = x / (z + xi + eps)
    y = y / (z + xi + eps)
    ro2 = x * x + y * y
    d = 1 + k1 * ro2 + k2 * ro2 * ro2
    x, y = x * d, y * d
    return P[0, 0] * x + P[0, 2], P[1, 1] * y + P[1, 2]


_LUT_CACHE = {}


def image2cam(norm, P, calib):
    """image2cam (:131-187): points [B,1,H,W,3] = LUT * norm, mask [B,1,H,W]."""
    B, _, H, W = norm.shape
    Xs, Ys, Zs, Ms = [], [], [], []
    for b in range(B):
        key = (H, W, P[b, 0, 0].item(), P[b, 1, 1].item(), P[b, 0, 2].item(), P[b, 1, 2].item(),
               calib[b]["distortion_parameters"]["k1"], calib[b]["distortion_parameters"]["k2"],
               calib[b]["mirror_parameters"]["xi"])
        if key not in _LUT_CACHE:
            _LUT_CACHE[key] = [torch.from_numpy(a)[None, None] for a in mei_lut(*key)]
        X, Y, Z, M = _LUT_CACHE[key]
        Xs.append(X); Ys.append(Y); Zs.append(Z); Ms.append(M)
    X, Y, Z, M = (torch.cat(v, 0) for v in (Xs, Ys, Zs, Ms))
    return torch.stack([X * norm, Y * norm, Z * norm], -1), M


def photometric_loss(outputs, inputs, frame_ids=(0, 1, -1), scales=(0, 1, 2, 3), noise=None):
    """FishEyeDecoder._generate_images_pred (monodepth2_decoder.py:355-411) + the unchanged
    compute_total_reprojection_loss / loss of the base class (:205-347), overlapped_mask=True."""
    target = inputs[("original_image", 0)]
    B, _, H, W = target.shape
    P, calib = inputs["P2"], inputs["calib_meta"]
    pm = inputs.get("patched_mask", torch.ones(B, H, W))
    losses = {}
    total = 0
    for scale in scales:
        norm = F.interpolate(outputs[("depth", scale, scale)], [H, W], mode="bilinear", align_corners=True)
        outputs[("depth", 0, scale)] = norm
        reproj = []
        for f in frame_ids[1:]:
            T = outputs[("cam_T_cam", f)]
            points, lmask = image2cam(norm, P, calib)                                 # [B,1,H,W,3]
            homo = torch.cat([points, torch.ones_like(points[..., :1])], -1).squeeze(1)[..., None]
            tp = torch.matmul(T[:, None, None], homo)[..., 0]                          # [B,H,W,4]
            uv = [cam2image(tp[b, ..., 0:3], P[b], calib[b]) for b in range(B)]
            u = torch.stack([a for a, _ in uv], 0)
            v = torch.stack([a for _, a in uv], 0)
            pix = torch.stack([u / max(W - 1, 1) * 2 - 1, v / max(H - 1, 1) * 2 - 1], -1)
            pred = F.grid_sample(inputs[("original_image", f)], pix, padding_mode="border", align_corners=True)
            outputs[("original_image", f, scale)] = pred
            pl = O.reprojection_loss(pred, target)
            rp = F.grid_sample((pm * lmask[:, 0]).unsqueeze(1).float(), pix, align_corners=True, mode="nearest")
            ov = (rp == 1)
            outputs[("overlapped_mask", f, scale)] = ov.squeeze(1)
            reproj.append(torch.where(ov, pl, torch.full_like(pl, 100.0)))
        reproj = torch.cat(reproj, 1)
        ident = torch.cat([O.reprojection_loss(inputs[("original_image", f)], target) for f in frame_ids[1:]], 1)
        if noise is not None:
            ident = ident + noise[scale]
        to_opt, idxs = torch.min(torch.cat((ident, reproj), 1), dim=1)
        outputs[("min_idx", scale)] = idxs
        to_opt = to_opt * pm
        loss = to_opt.sum() / (pm.sum() + 1e-6)
        disp = outputs[("disp", scale)]
        color = target if scale == 0 else F.adaptive_avg_pool2d(target, disp.shape[2:])
        mean_disp = disp.mean(2, True).mean(3, True)
        sm = O.smooth_loss(disp / (mean_disp + 1e-7), color) * 1e-5 / (2 ** scale)
        losses["smooth_loss/%d" % scale] = sm.detach()
        loss = loss + sm
        total = total + loss
        losses["loss/%d" % scale] = loss.detach()
    total = total / len(scales)
    losses["total_loss"] = total.detach()
    return total, losses


def get_prediction(norm, P, calib):
    """FishEyeDecoder.get_prediction (monodepth2_decoder.py:413-420): z = Z_lut * norm"""
    points, _ = image2cam(norm, P, calib)
    return points[..., 2]


def synthetic_calib(H, W, variant=0):
    """KITTI-360-like left/right fisheye parameters rescaled from the 1400x1400 sensor."""
    s = H / 1400.0
    if variant == 0:
        xi, k1, k2, g1, g2, u0, v0 = 2.2134047, 0.016798, 1.6548, 1336.3, 1335.8, 716.94, 705.76
    else:
        xi, k1, k2, g1, g2, u0, v0 = 2.5535139, 0.049134, 4.5014, 1485.4, 1484.9, 698.88, 698.14
    calib = {"distortion_parameters": {"k1": k1, "k2": k2}, "mirror_parameters": {"xi": xi}}
    P = torch.tensor([[g1 * s, 0, u0 * s * W / H, 0], [0, g2 * s, v0 * s, 0], [0, 0, 1, 0]], dtype=torch.float32)
    return P, calib
