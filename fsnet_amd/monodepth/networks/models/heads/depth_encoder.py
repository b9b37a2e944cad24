"""Depth decoder heads with the reference's constructor and state_dict names
(monodepth/networks/models/heads/depth_encoder.py:17-139), executed by the HIP engine."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from fsnet_amd.engine.nets import DepthDecoderRunner
from fsnet_amd.engine.runtime import RT, require_gpu
from fsnet_amd.vision_base.networks.blocks.blocks import ConvBnReLU
from fsnet_amd.vision_base.networks.models.backbone.resnet import nhwc_dense


class _DepthDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, nfeat, P2, *args):
        ctx.set_materialize_grads(False)
        feats = [nhwc_dense(f, f.dtype) for f in args[:nfeat]]
        outs, c = mod._runner.forward(feats, train=True, P2=P2)
        ctx.mod, ctx.c, ctx.nfeat, ctx.nparam = mod, c, nfeat, len(args) - nfeat
        mod._pending += 1
        if RT.dp is not None:
            RT.dp.note_forward(mod)
        flat = []
        for s in mod.scales:
            logits, depth, disp = outs[s][:3]
            flat += [logits.permute(0, 3, 1, 2)[:, : mod.num_output_channels], depth, disp]
            if mod._nout == 4:
                flat.append(outs[s][3])
        return tuple(flat)

    @staticmethod
    def backward(ctx, *g):
        mod = ctx.mod
        g_depth, g_disp, g_unc = {}, {}, {}
        no = mod._nout
        for k, s in enumerate(mod.scales):
            if g[no * k] is not None:
                raise NotImplementedError("gradient w.r.t. ('logits', s) is not supported by the HIP decoder")
            g_depth[s] = None if g[no * k + 1] is None else g[no * k + 1].contiguous().float()
            g_disp[s] = None if g[no * k + 2] is None else g[no * k + 2].contiguous().float()
            if no == 4:
                g_unc[s] = None if g[no * k + 3] is None else g[no * k + 3].contiguous().float()
        gfeats = mod._runner.backward(ctx.c, g_depth, g_disp, g_unc)
        ctx.c = None
        mod._pending -= 1
        if mod._pending == 0 and RT.dp is not None and not mod._runner.tail_pending:
            RT.dp.grads_ready(mod)      # (tail_pending: nets.flush_tail reduces the bucket behind the weight gradients)
        gf = tuple(None if t is None else t.permute(0, 3, 1, 2) for t in gfeats[: ctx.nfeat])
        return (None, None, None) + gf + (None,) * ctx.nparam


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, min_depth=0.1,
                 max_depth=100, base_fx=None):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = 'nearest'
        self.scales = list(scales)
        self.base_fx = base_fx
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.min_depth, self.max_depth = min_depth, max_depth
        self._build_depth_bins(min_depth, max_depth, num_output_channels)
        self._init_layers()
        self._nout = 4 if any(k[0] == "uncertain_logz" for k in self.convs) else 3   # tensors per scale
        self._runner = DepthDecoderRunner(self)
        self._pending = 0
        self._plist = None

    def _init_layers(self):
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = int(self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1])
            cout = int(self.num_ch_dec[i])
            self.convs[("upconv", i, 0)] = ConvBnReLU(cin, cout, kernel_size=(3, 3))
            cin = cout + (int(self.num_ch_enc[i - 1]) if (self.use_skips and i > 0) else 0)
            self.convs[("upconv", i, 1)] = ConvBnReLU(cin, cout, kernel_size=(3, 3), padding_mode='replicate')
        for s in self.scales:
            self.convs[("dispconv", s)] = nn.Conv2d(int(self.num_ch_dec[s]), self.num_output_channels, kernel_size=3,
                                                    padding=1, padding_mode='replicate')
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def _build_depth_bins(self, min_depth, max_depth, num_bins):
        lo, hi = np.log(min_depth), np.log(max_depth)
        self.register_buffer("depth_bins", torch.exp(torch.arange(lo, hi, (hi - lo) / num_bins)))

    sigmoid_head = True       # ('disp', s) = sigmoid(dispconv_s), depth by disp_to_depth (depth_encoder.py:90-111)

    def _check_head(self):
        if self.num_output_channels != 1:
            raise NotImplementedError("the sigmoid-disparity DepthDecoder has one output channel")

    def forward(self, input_features, P2=None):
        require_gpu(input_features[-1], type(self).__name__ + ".forward")
        self._check_head()
        feats = list(input_features)
        outputs = {}
        if torch.is_grad_enabled() and self.training:
            if self._plist is None:
                self._plist = list(self.parameters())
            flat = _DepthDecoderFn.apply(self, len(feats), P2, *feats, *self._plist)
            no = self._nout
            for k, s in enumerate(self.scales):
                outputs[('logits', s)], outputs[('depth', s, s)], outputs[('disp', s)] = flat[no * k: no * k + 3]
                if no == 4:
                    outputs[('uncertain_z', s)] = flat[no * k + 3]
            return outputs
        with torch.no_grad():
            outs, _ = self._runner.forward([nhwc_dense(f, f.dtype) for f in feats],
                                           train=self.decoder[0].sequence[1].training, P2=P2)
        for s in self.scales:
            logits, depth, disp = outs[s][:3]
            outputs[('logits', s)] = logits.permute(0, 3, 1, 2)[:, : self.num_output_channels]
            outputs[('depth', s, s)], outputs[('disp', s)] = depth, disp
            if self._nout == 4:
                outputs[('uncertain_z', s)] = outs[s][3]
        return outputs


class MultiChannelDepthDecoder(DepthDecoder):
    """softmax-over-log-spaced-depth-bins head (depth_encoder.py:114-139)."""
    sigmoid_head = False

    def _check_head(self):
        if self.num_output_channels not in (16, 32, 64):
            raise NotImplementedError("depth-bin head supports 16/32/64 bins")


class MultiChannelDepthDecoderUncertain(MultiChannelDepthDecoder):
    """MultiChannelDepthDecoder + a per-scale uncertainty head, sigmoid(3x3 replicate conv -> 1 channel)
    (depth_encoder.py:142-194; the reference version does not emit ('logits', s), this one keeps them)."""

    def _init_layers(self):
        super()._init_layers()
        for s in self.scales:
            self.convs[("uncertain_logz", s)] = nn.Conv2d(int(self.num_ch_dec[s]), 1, kernel_size=3, padding=1,
                                                          padding_mode='replicate')
        self.decoder = nn.ModuleList(list(self.convs.values()))       # upconvs, dispconvs, then uncertain_logz
