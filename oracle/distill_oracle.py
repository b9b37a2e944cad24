"""TEST INFRASTRUCTURE ONLY — CPU (torch fp32) restatement of the reference's self-distillation stage, SURVEY 8f
rank 3, on top of oracle/fsnet_oracle.py.  Only tests/ and tools/gen_golden.py import it.  PINNED: tests/golden/
distill.npz holds losses and gradient norms of the REAL DistillWPoseMeta (tools/gen_golden.py: gen_distill).

  MultiChannelDepthDecoderUncertain     monodepth/networks/models/heads/depth_encoder.py:142-194
  compute_distill_loss + loss           monodepth/networks/models/heads/monodepth2_decoder.py:185-203, 328-334
  MonoDepthInference                    monodepth/networks/models/meta_archs/teacher_model.py:5-32
  DistillWPoseMeta.forward_train        monodepth/networks/models/meta_archs/monodepth2_model.py:150-190
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import fsnet_oracle as O

UNC_BASE = 14       # decoder.14..17: uncertain_logz convs, after 10 upconvs and 4 dispconvs (depth_encoder.py:163-170)


def init_states(seed=0, teacher_seed=1, min_depth=0.5, max_depth=100.0):
    """(student+teacher) state dict with DistillWPoseMeta's key names"""
    sd = O.init_state(seed=seed, with_pose=False, min_depth=min_depth, max_depth=max_depth)
    rng = np.random.RandomState(seed + 1000)
    ch_dec = [16, 32, 64, 128, 256]
    for k, s in enumerate((0, 1, 2, 3)):
        fan = 1 * 3 * 3
        sd["head.depth_decoder.decoder.%d.weight" % (UNC_BASE + k)] = torch.from_numpy(
            (rng.randn(1, ch_dec[s], 3, 3) * math.sqrt(2.0 / fan) * 0.02).astype(np.float32))
        sd["head.depth_decoder.decoder.%d.bias" % (UNC_BASE + k)] = torch.from_numpy((0.3 + 0.02 * rng.randn(1)).astype(np.float32))
    t = O.init_state(seed=teacher_seed, with_pose=False, min_depth=min_depth, max_depth=max_depth)
    rngt = np.random.RandomState(teacher_seed + 2000)
    for k, v in t.items():
        name = k.replace("head.depth_decoder.", "depth_decoder.")
        if name.endswith("running_mean"):          # a trained teacher: non-trivial running statistics (eval-mode BN)
            v = torch.from_numpy((0.1 * rngt.randn(*v.shape)).astype(np.float32))
        elif name.endswith("running_var"):
            v = torch.from_numpy((1.0 + 0.3 * rngt.rand(*v.shape)).astype(np.float32))
        sd["teacher_net." + name] = v
    return sd


def is_student_param(name):
    return O.is_param(name) and not name.startswith("teacher_net.")


def decoder_uncertain_forward(sd, prefix, feats, min_depth, max_depth, scales=(0, 1, 2, 3), train=True):
    """depth_encoder.py:172-194: MultiChannelDepthDecoder.forward + ('uncertain_z', s) = sigmoid(uncertain_logz(x))"""
    out = {}
    x = feats[-1]
    idx = 0
    for i in range(4, -1, -1):
        for j in range(2):
            p = "%sdecoder.%d.sequence." % (prefix, idx)
            if j == 1:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
                if i > 0:
                    x = torch.cat([x, feats[i - 1]], 1)
            x = O._conv_pad(x, sd[p + "0.weight"], sd[p + "0.bias"], "replicate" if j == 1 else "zeros")
            x = F.relu(O._bn(sd, p + "1", x, train))
            idx += 1
        if i in scales:
            k = list(scales).index(i)
            logits = O._conv_pad(x, sd["%sdecoder.%d.weight" % (prefix, 10 + k)], sd["%sdecoder.%d.bias" % (prefix, 10 + k)],
                                 "replicate")
            depth = O.gather_activation(logits, sd[prefix + "depth_bins"])
            out[("depth", i, i)] = depth
            out[("disp", i)] = O.depth_to_disp(depth, min_depth, max_depth)
            ul = O._conv_pad(x, sd["%sdecoder.%d.weight" % (prefix, UNC_BASE + k)],
                             sd["%sdecoder.%d.bias" % (prefix, UNC_BASE + k)], "replicate")
            out[("uncertain_z", i)] = torch.sigmoid(ul)
    return out


def distill_loss(pred, teacher, uncertain=None):
    """monodepth2_decoder.py:185-203 with is_unscaled_distill=False"""
    error = (teacher.detach() - pred).abs()
    loss = error / uncertain + torch.log(uncertain + 1e-5) if uncertain is not None else error
    return loss.mean()


def forward_train(sd, data, min_depth=0.5, max_depth=100.0, frame_ids=(0, 1, -1), scales=(0, 1, 2, 3),
                  distillation_loss_weight=0.3, is_uncertain_distill=True, noise=None):
    feats = O.resnet_forward(sd, "depth_backbone.", data[("image", 0)], 18)
    outputs = decoder_uncertain_forward(sd, "head.depth_decoder.", feats, min_depth, max_depth, scales)
    with torch.no_grad():                                                       # teacher: eval mode, detached
        tf = O.resnet_forward(sd, "teacher_net.depth_backbone.", data[("image", 0)], 18, train=False)
        to = O.depth_decoder_forward(sd, "teacher_net.depth_decoder.", tf, min_depth, max_depth, scales, train=False)
    for s in scales:
        outputs[("teacher_depth", s, s)] = to[("depth", s, s)]
    for f in frame_ids[1:]:
        outputs[("cam_T_cam", f)] = data[("relative_pose", f)]
    total, losses = O.photometric_loss(outputs, data, frame_ids, scales, True, noise)
    for s in scales:
        dl = distill_loss(outputs[("depth", s, s)], outputs[("teacher_depth", s, s)],
                          outputs[("uncertain_z", s)] if is_uncertain_distill else None)
        losses["distilation/%d" % s] = dl.detach()
        total = total + dl * distillation_loss_weight
    losses["total_loss"] = total.detach()
    return total, losses, outputs
