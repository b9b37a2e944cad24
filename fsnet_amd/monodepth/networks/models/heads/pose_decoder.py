"""Pose decoder with the reference's constructor / state_dict names
(monodepth/networks/models/heads/pose_decoder.py:5-45), executed by the HIP engine; the spatial
mean, the 0.01 scaling and the axis-angle -> 4x4 transform are one fused kernel."""
from collections import OrderedDict

import torch
import torch.nn as nn

from fsnet_amd.engine.nets import PoseDecoderRunner
from fsnet_amd.engine.runtime import RT, require_gpu
from fsnet_amd.vision_base.networks.models.backbone.resnet import nhwc_dense


class _PoseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, feat, invert, *params):
        ctx.set_materialize_grads(False)
        aa, tr, T, c = mod._runner.forward(nhwc_dense(feat, feat.dtype), invert)
        ctx.mod, ctx.c, ctx.nparam = mod, c, len(params)
        mod._pending += 1
        if RT.dp is not None:
            RT.dp.note_forward(mod)
        return aa, tr, T

    @staticmethod
    def backward(ctx, g_aa, g_tr, g_T):
        mod = ctx.mod
        if g_aa is not None or g_tr is not None:
            raise NotImplementedError("gradients w.r.t. axisangle/translation outputs (pose_loss_weight > 0) "
                                      "are not supported by the HIP pose decoder yet")
        if g_T is None:
            g_T = torch.zeros(ctx.c["x3"].shape[0], 4, 4, device=ctx.c["x3"].device)
        d = mod._runner.backward(ctx.c, g_T.contiguous().float())
        ctx.c = None
        mod._pending -= 1
        if mod._pending == 0 and RT.dp is not None:
            RT.dp.grads_ready(mod)
        return (None, d.permute(0, 3, 1, 2), None) + (None,) * ctx.nparam


class _PosePairsFn(torch.autograd.Function):
    """G stacked calls of the pose decoder (one per image pair) as one pass over the stacked encoder feature."""

    @staticmethod
    def forward(ctx, mod, feat, inverts, *params):
        ctx.set_materialize_grads(False)
        aa, tr, T, c = mod._runner.forward(nhwc_dense(feat, feat.dtype), inverts)
        ctx.mod, ctx.c, ctx.nparam, ctx.G = mod, c, len(params), len(inverts)
        mod._pending += 1
        if RT.dp is not None:
            RT.dp.note_forward(mod)
        return tuple(aa) + tuple(tr) + tuple(T)

    @staticmethod
    def backward(ctx, *g):
        mod, G = ctx.mod, ctx.G
        if any(gi is not None for gi in g[:2 * G]):
            raise NotImplementedError("gradients w.r.t. axisangle/translation outputs (pose_loss_weight > 0) "
                                      "are not supported by the HIP pose decoder yet")
        x3 = ctx.c["x3"]
        B = x3.shape[0] // G
        dT = [gi.contiguous().float() if gi is not None else torch.zeros(B, 4, 4, device=x3.device) for gi in g[2 * G:]]
        d = mod._runner.backward(ctx.c, dT)
        ctx.c = None
        mod._pending -= 1
        if mod._pending == 0 and RT.dp is not None:
            RT.dp.grads_ready(mod)
        return (None, d.permute(0, 3, 1, 2), None) + (None,) * ctx.nparam


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        if num_input_features != 1 or stride != 1:
            raise NotImplementedError("HIP PoseDecoder supports num_input_features=1, stride=1 (FSNet configs)")
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(int(self.num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))
        self._runner = PoseDecoderRunner(self)
        self._pending = 0
        self._plist = None

    def forward_with_transform(self, input_features, invert):
        """-> (axisangle [B,F,1,3], translation [B,F,1,3], T [B,4,4] of frame 0)."""
        feat = input_features[0][-1]
        require_gpu(feat, "PoseDecoder")
        if torch.is_grad_enabled() and self.training:
            if self._plist is None:
                self._plist = list(self.parameters())
            return _PoseFn.apply(self, feat, bool(invert), *self._plist)
        with torch.no_grad():
            aa, tr, T, _ = self._runner.forward(nhwc_dense(feat, feat.dtype), bool(invert))
        return aa, tr, T

    def forward_pairs_with_transform(self, input_features, inverts):
        """input_features[0][-1] stacks G calls along N -> [(axisangle, translation, T) per call]."""
        feat = input_features[0][-1]
        require_gpu(feat, "PoseDecoder")
        inverts = tuple(bool(i) for i in inverts)
        G = len(inverts)
        if torch.is_grad_enabled() and self.training:
            if self._plist is None:
                self._plist = list(self.parameters())
            o = _PosePairsFn.apply(self, feat, inverts, *self._plist)
        else:
            with torch.no_grad():
                aa, tr, T, _ = self._runner.forward(nhwc_dense(feat, feat.dtype), inverts)
            o = tuple(aa) + tuple(tr) + tuple(T)
        return [(o[g], o[G + g], o[2 * G + g]) for g in range(G)]

    def forward(self, input_features):
        aa, tr, _ = self.forward_with_transform(input_features, False)
        return aa, tr
