"""CPU oracle for FSNet's self-supervised monodepth training step.

THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path (fsnet_amd/) never does.

It restates, in plain fp32 torch-CPU ops (float64 exactly where the reference promotes), the
algorithm of the reference hot path.  Every function cites the reference file:line it follows.
Parity pin: tools/gen_golden.py imports the real reference in the build container, runs it on
seeded synthetic inputs and stores inputs + reference outputs under tests/golden/*.npz;
tests/test_oracle_golden.py checks this restatement against those vectors.

State is a flat dict {reference state_dict key: tensor} (SURVEY §8b names), so the same
checkpoint drives the reference, this oracle and the HIP implementation.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------------
# parameters: seeded, framework-independent initialisation (shapes follow the reference modules)
# ----------------------------------------------------------------------------------------------
def resnet_param_shapes(prefix, depth=18, num_input_images=1):
    """vision_base/networks/models/backbone/resnet.py:21-50 (BasicBlock), 53-89 (Bottleneck),
    119-123 (stem), 136-151 (_make_layer), 270-284 (depth -> block counts)."""
    blocks = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3])}[depth]
    kind, layers = blocks
    exp = 1 if kind == "basic" else 4
    shapes = []

    def bn(name, c):
        shapes.extend([(name + ".weight", (c,)), (name + ".bias", (c,)), (name + ".running_mean", (c,)),
                       (name + ".running_var", (c,)), (name + ".num_batches_tracked", ())])

    shapes.append((prefix + "conv1.weight", (64, 3 * num_input_images, 7, 7)))
    bn(prefix + "bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers)):
        stride = 1 if li == 0 else 2
        for b in range(nblk):
            p = "%slayer%d.%d." % (prefix, li + 1, b)
            s = stride if b == 0 else 1
            if kind == "basic":
                shapes.append((p + "conv1.weight", (planes, inplanes, 3, 3))); bn(p + "bn1", planes)
                shapes.append((p + "conv2.weight", (planes, planes, 3, 3))); bn(p + "bn2", planes)
            else:
                shapes.append((p + "conv1.weight", (planes, inplanes, 1, 1))); bn(p + "bn1", planes)
                shapes.append((p + "conv2.weight", (planes, planes, 3, 3))); bn(p + "bn2", planes)
                shapes.append((p + "conv3.weight", (planes * 4, planes, 1, 1))); bn(p + "bn3", planes * 4)
            if b == 0 and (s != 1 or inplanes != planes * exp):
                shapes.append((p + "downsample.0.weight", (planes * exp, inplanes, 1, 1)))
                bn(p + "downsample.1", planes * exp)
            inplanes = planes * exp
    return shapes


def num_ch_enc(depth):
    return [64, 64, 128, 256, 512] if depth in (18, 34) else [64, 256, 512, 1024, 2048]


def depth_decoder_param_shapes(prefix, ch_enc, num_out=16, scales=(0, 1, 2, 3)):
    """monodepth/networks/models/heads/depth_encoder.py:45-66 (ModuleList order: upconv(4,0),
    upconv(4,1), ..., upconv(0,1), then dispconv per scale); blocks.py:41-46 (ConvBnReLU)."""
    ch_dec = [16, 32, 64, 128, 256]
    shapes = []
    idx = 0
    for i in range(4, -1, -1):
        cin = ch_enc[-1] if i == 4 else ch_dec[i + 1]
        for j in range(2):
            if j == 1:
                cin = ch_dec[i] + (ch_enc[i - 1] if i > 0 else 0)
            p = "%sdecoder.%d.sequence." % (prefix, idx)
            shapes.append((p + "0.weight", (ch_dec[i], cin, 3, 3)))
            shapes.append((p + "0.bias", (ch_dec[i],)))
            for nm in ("weight", "bias", "running_mean", "running_var"):
                shapes.append((p + "1." + nm, (ch_dec[i],)))
            shapes.append((p + "1.num_batches_tracked", ()))
            idx += 1
    for s in scales:
        shapes.append(("%sdecoder.%d.weight" % (prefix, idx), (num_out, ch_dec[s], 3, 3)))
        shapes.append(("%sdecoder.%d.bias" % (prefix, idx), (num_out,)))
        idx += 1
    return shapes


def pose_decoder_param_shapes(prefix, ch_last=512, num_input_features=1, num_frames=2):
    """monodepth/networks/models/heads/pose_decoder.py:16-24."""
    return [
        (prefix + "net.0.weight", (256, ch_last, 1, 1)), (prefix + "net.0.bias", (256,)),
        (prefix + "net.1.weight", (256, num_input_features * 256, 3, 3)), (prefix + "net.1.bias", (256,)),
        (prefix + "net.2.weight", (256, 256, 3, 3)), (prefix + "net.2.bias", (256,)),
        (prefix + "net.3.weight", (6 * num_frames, 256, 1, 1)), (prefix + "net.3.bias", (6 * num_frames,)),
    ]


def depth_bins(min_depth, max_depth, num_bins):
    """depth_encoder.py:68-74 (_build_depth_bins: log-spaced, float32 arange)."""
    lo, hi = np.log(min_depth), np.log(max_depth)
    return torch.exp(torch.arange(lo, hi, (hi - lo) / num_bins))


def model_param_shapes(depth=18, with_pose=True, num_out=16, scales=(0, 1, 2, 3)):
    shapes = resnet_param_shapes("depth_backbone.", depth, 1)
    if with_pose:
        shapes += resnet_param_shapes("pose_backbone.", depth, 2)
    shapes += depth_decoder_param_shapes("head.depth_decoder.", num_ch_enc(depth), num_out, scales)
    if with_pose:
        shapes += pose_decoder_param_shapes("head.pose_decoder.", num_ch_enc(depth)[-1])
    return shapes


def init_state(seed=0, depth=18, with_pose=True, num_out=16, min_depth=0.5, max_depth=100.0,
               scales=(0, 1, 2, 3), gain=1.0):
    """Deterministic numpy-seeded state dict with the reference's key names.  Conv weights are
    He-normal (fan-out, as resnet.py:125-131), BN affine perturbed away from (1, 0) so that the
    parity tests see non-trivial scale/shift, running stats at their initial (0, 1)."""
    rng = np.random.RandomState(seed)
    sd = {}
    for name, shape in model_param_shapes(depth, with_pose, num_out, scales):
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif len(shape) == 4:
            fan = shape[0] * shape[2] * shape[3]
            sd[name] = torch.from_numpy((rng.randn(*shape) * gain * math.sqrt(2.0 / fan)).astype(np.float32))
        elif ".bn" in name or "downsample.1" in name or "sequence.1" in name:
            if name.endswith("weight"):
                sd[name] = torch.from_numpy((1.0 + 0.1 * rng.randn(*shape)).astype(np.float32))
            else:
                sd[name] = torch.from_numpy((0.05 * rng.randn(*shape)).astype(np.float32))
        else:  # conv bias
            sd[name] = torch.from_numpy((0.02 * rng.randn(*shape)).astype(np.float32))
    sd["head.depth_decoder.depth_bins"] = depth_bins(min_depth, max_depth, num_out)
    return sd


def is_param(name):
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked") or name.endswith("depth_bins"))


# ----------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------
def _bn(sd, name, x, train=True):
    """nn.BatchNorm2d in train mode (resnet.py:169-175 with norm_eval=False): batch statistics,
    running stats updated in place with momentum 0.1 / unbiased variance."""
    if train:
        sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def frozen_names(prefix, frozen_stages):
    """parameter-name prefixes that ResNet.freeze_stages (resnet.py:179-193) takes out of training"""
    out = []
    if frozen_stages >= 0:
        out += [prefix + "conv1.", prefix + "bn1."]
    out += ["%slayer%d." % (prefix, i) for i in range(1, frozen_stages + 1)]
    return tuple(out)


def resnet_forward(sd, prefix, x, depth=18, train=True, frozen_stages=-1, norm_eval=False):
    """resnet.py:199-213 (forward, out_indices=(-1,0,1,2,3)), 33-50 (BasicBlock), 70-89
    (Bottleneck, stride on the 3x3).  frozen_stages / norm_eval: ResNet.train() (:169-197) leaves the BatchNorms of
    the frozen stem / stages — or all of them — in eval mode inside a training step."""
    kind, layers = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3])}[depth]
    outs = []
    glob_train = train
    train = glob_train and not norm_eval and frozen_stages < 0
    x = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, prefix + "bn1", x, train))
    outs.append(x)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblk in enumerate(layers):
        train = glob_train and not norm_eval and (li + 1) > frozen_stages
        for b in range(nblk):
            p = "%slayer%d.%d." % (prefix, li + 1, b)
            stride = 2 if (li > 0 and b == 0) else 1
            res = x
            if kind == "basic":
                out = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, p + "bn1", out, train))
                out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
                out = _bn(sd, p + "bn2", out, train)
            else:
                out = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"]), train))
                out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, p + "bn2", out, train))
                out = _bn(sd, p + "bn3", F.conv2d(out, sd[p + "conv3.weight"]), train)
            if (p + "downsample.0.weight") in sd:
                res = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride)
                res = _bn(sd, p + "downsample.1", res, train)
            x = F.relu(out + res)
        outs.append(x)
    return outs


def _conv_pad(x, w, b, mode):
    """3x3 conv, padding 1, zeros or replicate padding (depth_encoder.py:52,59,62)."""
    if mode == "replicate":
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w, b)
    return F.conv2d(x, w, b, padding=1)


def depth_to_disp(depth, min_depth, max_depth):
    """monodepth_utils.py:19-24."""
    return (1 / depth - 1 / max_depth) / (1 / min_depth - 1 / max_depth)


def gather_activation(logits, bins):
    """depth_encoder.py:76-88 (_gather_activation): clamp +-10, softmax over bins, expectation."""
    x = torch.clamp(logits, -10.0, 10.0)
    act = torch.softmax(x, dim=1)
    return torch.sum(act * bins.reshape(1, -1, 1, 1), dim=1, keepdim=True)


def depth_decoder_forward(sd, prefix, feats, min_depth, max_depth, scales=(0, 1, 2, 3), train=True, P2=None,
                          base_fx=None, sigmoid=False):
    """depth_encoder.py:119-139 (MultiChannelDepthDecoder.forward / gather_output); with base_fx and P2 the
    focal-length scale of _get_scale (:36-43): depth *= fx / base_fx, disparity against the scaled range.
    sigmoid=True: the base class DepthDecoder.forward (:90-111) — disp = sigmoid(dispconv), depth through
    disp_to_depth (monodepth_utils.py:8-17) times the focal-length scale."""
    out = {}
    depth_scale = 1
    if base_fx is not None and P2 is not None:
        depth_scale = (P2[:, 0, 0] / base_fx).reshape([-1, 1, 1, 1])
    x = feats[-1]
    idx = 0
    disp_base = 10
    for i in range(4, -1, -1):
        for j in range(2):
            p = "%sdecoder.%d.sequence." % (prefix, idx)
            if j == 1:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
                if i > 0:
                    x = torch.cat([x, feats[i - 1]], 1)
            x = _conv_pad(x, sd[p + "0.weight"], sd[p + "0.bias"], "replicate" if j == 1 else "zeros")
            x = F.relu(_bn(sd, p + "1", x, train))
            idx += 1
        if i in scales:
            k = disp_base + list(scales).index(i)
            logits = _conv_pad(x, sd["%sdecoder.%d.weight" % (prefix, k)], sd["%sdecoder.%d.bias" % (prefix, k)],
                               "replicate")
            out[("logits", i)] = logits
            if sigmoid:
                disp = torch.sigmoid(logits)
                scaled = 1 / max_depth + (1 / min_depth - 1 / max_depth) * disp
                out[("disp", i)] = disp
                out[("depth", i, i)] = (1 / scaled) * depth_scale
                continue
            depth = gather_activation(logits, sd[prefix + "depth_bins"])
            if base_fx is not None:
                depth = depth * depth_scale
            out[("depth", i, i)] = depth
            out[("disp", i)] = depth_to_disp(depth, min_depth * depth_scale, max_depth * depth_scale)
    return out


def pose_decoder_forward(sd, prefix, last_feat, num_frames=2):
    """pose_decoder.py:26-45."""
    x = F.relu(F.conv2d(last_feat, sd[prefix + "net.0.weight"], sd[prefix + "net.0.bias"]))
    x = F.relu(F.conv2d(x, sd[prefix + "net.1.weight"], sd[prefix + "net.1.bias"], padding=1))
    x = F.relu(F.conv2d(x, sd[prefix + "net.2.weight"], sd[prefix + "net.2.bias"], padding=1))
    x = F.conv2d(x, sd[prefix + "net.3.weight"], sd[prefix + "net.3.bias"])
    x = x.mean(3).mean(2)
    x = 0.01 * x.view(-1, num_frames, 1, 6)
    return x[..., :3], x[..., 3:]


def rot_from_axisangle(vec):
    """monodepth_utils.py:298-337 (Rodrigues; axis = v / (|v| + 1e-7))."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    Cc = 1 - ca
    x, y, z = axis[..., 0:1], axis[..., 1:2], axis[..., 2:3]
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * Cc, y * Cc, z * Cc
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    B = vec.shape[0]
    rows = [
        torch.cat([x * xC + ca, xyC - zs, zxC + ys], 2),
        torch.cat([xyC + zs, y * yC + ca, yzC - xs], 2),
        torch.cat([zxC - ys, yzC + xs, z * zC + ca], 2),
    ]
    R3 = torch.cat(rows, 1)  # [B,3,3]
    R = torch.zeros(B, 4, 4)
    R = torch.cat([torch.cat([R3, torch.zeros(B, 3, 1)], 2), torch.tensor([0., 0., 0., 1.]).expand(B, 1, 4)], 1)
    return R


def transformation_from_parameters(axisangle, translation, invert=False):
    """monodepth_utils.py:31-63."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    B = t.shape[0]
    T = torch.eye(4).expand(B, 4, 4).clone()
    T = torch.cat([torch.cat([torch.eye(3).expand(B, 3, 3), t.contiguous().view(-1, 3, 1)], 2),
                   torch.tensor([0., 0., 0., 1.]).expand(B, 1, 4)], 1)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


# ----------------------------------------------------------------------------------------------
# photometric loss chain
# ----------------------------------------------------------------------------------------------
def intrinsics(P2):
    """monodepth2_decoder.py:82-85: K (4x4, float64 numpy) from P2[:, :3, :3]; pinv in float64;
    both cast to fp32 at use (88, 90)."""
    B = P2.shape[0]
    K = np.zeros([B, 4, 4])
    K[:, 0:3, 0:3] = P2[:, 0:3, 0:3].double().numpy()
    K[:, 3, 3] = 1
    inv_K = np.linalg.pinv(K)
    return torch.from_numpy(K).float(), torch.from_numpy(inv_K).float()


def backproject(depth, inv_K):
    """monodepth_utils.py:105-117,132-143: pixel grid (x = col, y = row, 1), cam = depth * K^-1 p."""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], 0).unsqueeze(0).repeat(B, 1, 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, torch.ones(B, 1, H * W)], 1)


def project(points, K, T, H, W, eps=1e-7):
    """monodepth_utils.py:154-165: P = (K T)[:3]; perspective divide with +eps; normalise to [-1,1]."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    pix = torch.stack([pix[..., 0] / (W - 1), pix[..., 1] / (H - 1)], -1)
    return (pix - 0.5) * 2


def ssim(x, y):
    """monodepth_utils.py:184-215: 3x3 mean filters on reflection-padded images, C1=1e-4, C2=9e-4."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred, target, ssim_weight=0.85):
    """monodepth2_decoder.py:118-128."""
    l1 = torch.abs(target - pred).mean(1, True)
    return ssim_weight * ssim(pred, target).mean(1, True) + (1 - ssim_weight) * l1


def smooth_loss(disp, img):
    """monodepth_utils.py:168-181."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def photometric_loss(outputs, inputs, frame_ids=(0, 1, -1), scales=(0, 1, 2, 3), overlapped_mask=True,
                     noise=None):
    """monodepth2_decoder.py:61-116 (_generate_images_pred) + 205-304
    (compute_total_reprojection_loss) + 306-347 (loss) with the shipped KITTI options
    (no motion_mask / ssim_weight / pose / distill terms).  `noise`: optional dict scale ->
    [B,2,H,W] tie-break noise replacing torch.randn*1e-5 (258-259); None = zeros."""
    target = inputs[("original_image", 0)]
    B, _, H, W = target.shape
    K, inv_K = intrinsics(inputs["P2"])
    pm = inputs.get("patched_mask", torch.ones(B, H, W))
    losses = {}
    total = 0
    for scale in scales:
        depth = F.interpolate(outputs[("depth", scale, scale)], [H, W], mode="bilinear", align_corners=True)
        outputs[("depth", 0, scale)] = depth
        reproj = []
        for f in frame_ids[1:]:
            T = outputs[("cam_T_cam", f)]
            cam = backproject(depth, inv_K)
            pix = project(cam, K, T, H, W)
            pred = F.grid_sample(inputs[("original_image", f)], pix, padding_mode="border", align_corners=True)
            outputs[("original_image", f, scale)] = pred
            pl = reprojection_loss(pred, target)
            if overlapped_mask:
                rp = F.grid_sample(pm.unsqueeze(1).float(), pix, align_corners=True, mode="nearest")
                ov = (rp == 1)
                outputs[("overlapped_mask", f, scale)] = ov.squeeze(1)
                pl = torch.where(ov, pl, torch.full_like(pl, 100.0))  # 231-235 (blocks the gradient)
            reproj.append(pl)
        reproj = torch.cat(reproj, 1)
        if "motion_mask" in inputs:
            # precomputed motion mask instead of the identity auto-mask (monodepth2_decoder.py:243-246): the value is
            # unchanged, the gradient is blocked where the mask is set
            mm = inputs["motion_mask"]
            to_opt, idxs = torch.min(reproj, dim=1)
            to_opt = to_opt.detach() * mm + to_opt * (1 - mm)
            idxs = idxs + 2
        else:
            ident = torch.cat([reprojection_loss(inputs[("original_image", f)], target) for f in frame_ids[1:]], 1)
            if noise is not None:
                ident = ident + noise[scale]
            combined = torch.cat((ident, reproj), dim=1)
            to_opt, idxs = torch.min(combined, dim=1)
        outputs[("min_idx", scale)] = idxs
        to_opt = to_opt * pm  # float64 promotion when patched_mask is float64 (270-272)
        loss = to_opt.sum() / (pm.sum() + 1e-6)
        disp = outputs[("disp", scale)]
        color = target if scale == 0 else F.adaptive_avg_pool2d(target, disp.shape[2:])
        mean_disp = disp.mean(2, True).mean(3, True)
        sm = smooth_loss(disp / (mean_disp + 1e-7), color) * 1e-5 / (2 ** scale)
        losses["smooth_loss/%d" % scale] = sm.detach()
        loss = loss + sm
        total = total + loss
        losses["loss/%d" % scale] = loss.detach()
    total = total / len(scales)
    losses["total_loss"] = total.detach()
    return total, losses


# ----------------------------------------------------------------------------------------------
# meta-arch forward and one optimisation step
# ----------------------------------------------------------------------------------------------
def forward_train(sd, data, depth=18, with_pose=True, min_depth=0.5, max_depth=100.0,
                  frame_ids=(0, 1, -1), scales=(0, 1, 2, 3), overlapped_mask=True, noise=None, base_fx=None,
                  frozen_stages=-1, norm_eval=False):
    """MonoDepthMeta.forward_train (monodepth2_model.py:24-46) when with_pose, else
    MonoDepthWPose.forward_train (85-130, dataset poses, no residual pose net)."""
    feats = resnet_forward(sd, "depth_backbone.", data[("image", 0)], depth, frozen_stages=frozen_stages,
                           norm_eval=norm_eval)
    # MonoDepthWPose hands P2 to the decoder only when base_fx is set (monodepth2_model.py:91)
    outputs = depth_decoder_forward(sd, "head.depth_decoder.", feats, min_depth, max_depth, scales,
                                    P2=(data["P2"] if base_fx is not None else None), base_fx=base_fx)
    for f in frame_ids[1:]:
        if with_pose:
            pair = [data[("image", f)], data[("image", 0)]] if f < 0 else [data[("image", 0)], data[("image", f)]]
            pf = resnet_forward(sd, "pose_backbone.", torch.cat(pair, 1), depth, frozen_stages=frozen_stages,
                                norm_eval=norm_eval)
            aa, tr = pose_decoder_forward(sd, "head.pose_decoder.", pf[-1])
            outputs[("axisangle", f)], outputs[("translation", f)] = aa, tr
            outputs[("cam_T_cam", f)] = transformation_from_parameters(aa[:, 0], tr[:, 0], invert=(f < 0))
        else:
            outputs[("cam_T_cam", f)] = data[("relative_pose", f)]
    total, losses = photometric_loss(outputs, data, frame_ids, scales, overlapped_mask, noise)
    return total, losses, outputs


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as used at base_training_hooks.py:46-47: joint L2 norm,
    coefficient max_norm / (norm + 1e-6) clamped to 1."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    return norm, [g * coef for g in grads]


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (optimizers.py:7-8), default betas/eps, L2 weight decay added to grad."""
    if weight_decay != 0:
        g = g + weight_decay * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class OracleTrainer:
    """One full optimisation step = base_training_hooks.py:22-49 (zero_grad, forward,
    loss.mean().backward(), clip_grad_norm_, Adam.step)."""

    def __init__(self, sd, depth=18, with_pose=True, lr=1e-4, clip=35.0, min_depth=0.5, max_depth=100.0,
                 weight_decay=0.0, base_fx=None, frozen_stages=-1, norm_eval=False):
        self.base_fx = base_fx
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        # torch.optim.Adam never touches a parameter without a gradient (requires_grad False after freeze_stages)
        self.frozen = frozen_names("depth_backbone.", frozen_stages) + frozen_names("pose_backbone.", frozen_stages)
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = [k for k in self.sd if is_param(k)]
        self.depth, self.with_pose, self.lr, self.clip = depth, with_pose, lr, clip
        self.min_depth, self.max_depth, self.wd = min_depth, max_depth, weight_decay
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.t = 0

    def step(self, data, noise=None):
        for k in self.names:
            self.sd[k] = self.sd[k].detach().requires_grad_(True)
        total, losses, outputs = forward_train(self.sd, data, self.depth, self.with_pose, self.min_depth,
                                               self.max_depth, noise=noise, base_fx=self.base_fx,
                                               frozen_stages=self.frozen_stages, norm_eval=self.norm_eval)
        grads = torch.autograd.grad(total.mean(), [self.sd[k] for k in self.names], allow_unused=True)
        grads = [g if (g is not None and not k.startswith(self.frozen)) else torch.zeros_like(self.sd[k])
                 for g, k in zip(grads, self.names)]
        raw = dict(zip(self.names, grads))
        norm = None
        if self.clip is not None:
            norm, grads = clip_grad_norm(grads, self.clip)
        self.t += 1
        with torch.no_grad():
            for k, g in zip(self.names, grads):
                p = self.sd[k].detach()
                if k.startswith(self.frozen):
                    self.sd[k] = p
                    continue
                adam_step(p, g, self.m[k], self.v[k], self.t, self.lr, weight_decay=self.wd)
                self.sd[k] = p
        return total.detach(), losses, outputs, raw, norm


# ----------------------------------------------------------------------------------------------
# synthetic batches: smooth textured frames, KITTI-like intrinsics, unit poses
# ----------------------------------------------------------------------------------------------
def synthetic_batch(B, H, W, seed=0, frame_ids=(0, 1, -1)):
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(0, 1, H).view(1, 1, H, 1)
    xs = torch.linspace(0, 1, W).view(1, 1, 1, W)
    data = {}
    ph = torch.rand(B, 3, 1, 1, generator=g) * 6.28
    fr = 3 + torch.rand(B, 3, 1, 1, generator=g) * 9
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    for f in frame_ids:
        sh = 3.0 * f / W
        img = 0.5 + 0.25 * torch.sin(fr * 6.28 * (xs + sh) + ph) * torch.cos(fr * 3.1 * ys + 0.5 * ph) \
            + 0.2 * torch.sin(37.0 * (xs + sh) * ys + ph) + 0.03 * torch.rand(B, 3, H, W, generator=g)
        img = img.clamp(0, 1).float()
        data[("original_image", f)] = img
        data[("image", f)] = ((img - mean) / std).float()
    P2 = torch.zeros(B, 3, 4)
    P2[:, 0, 0] = 0.58 * W; P2[:, 0, 2] = 0.5 * W
    P2[:, 1, 1] = 1.92 * H; P2[:, 1, 2] = 0.5 * H
    P2[:, 2, 2] = 1
    data["P2"] = P2
    for f in frame_ids[1:]:
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, 0, 3] = 0.01
        T[:, 2, 3] = -0.8 if f > 0 else 0.8
        data[("relative_pose", f)] = T
    data["patched_mask"] = torch.ones(B, H, W, dtype=torch.float64)
    return data
