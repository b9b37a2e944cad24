"""ResNet encoder with the reference's constructor, state_dict names and 5-level feature pyramid
(reference vision_base/networks/models/backbone/resnet.py:21-213, 270-284), executed by the HIP engine.

The nn.Conv2d / nn.BatchNorm2d children are parameter containers only.  forward() takes the NCHW fp32
batch the data layer provides, converts it once to NHWC (compute dtype) and returns the five features as
channels-last tensors (logical NCHW), connected to autograd through ONE custom Function."""
import math

import torch
import torch.nn as nn

from fsnet_amd.engine.nets import EncoderPass, ResNetRunner
from fsnet_amd.engine.runtime import RT, require_gpu
from fsnet_amd.hip import ops


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


def nhwc_dense(t, dtype):
    """logical-NCHW tensor -> dense NHWC view/copy in `dtype` (no copy for channels-last inputs)."""
    v = t.permute(0, 2, 3, 1)
    if v.dtype != dtype:
        v = v.to(dtype)
    return v if v.is_contiguous() else v.contiguous()


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, groups, *params):
        ctx.set_materialize_grads(False)
        feats, c = mod._runner.forward(x, train=True, groups=groups)
        ctx.mod, ctx.c, ctx.dtype = mod, c, x.dtype
        mod._pending += 1
        if RT.dp is not None:
            RT.dp.note_forward(mod)
        return tuple(f.permute(0, 3, 1, 2) for f in feats)

    @staticmethod
    def backward(ctx, *g):
        mod = ctx.mod
        gf = [None if gi is None else nhwc_dense(gi, ctx.dtype) for gi in g]
        mod._runner.backward(ctx.c, gf)
        ctx.c = None
        mod._pending -= 1
        if mod._pending == 0 and RT.dp is not None:
            RT.dp.grads_ready(mod)
        return (None, None, None) + (None,) * len(mod._plist)


class _ResNetLanesFn(torch.autograd.Function):
    """Two encoders as the lanes of ONE pass (engine/nets.py, EncoderPass): the autograd node of the depth encoder's call
    and the stacked pose encoder's call of a training step (monodepth2_model.py:24-43) — every launch of the pass serves
    both networks.  Outputs: the first lane's five features, then the second's."""

    @staticmethod
    def forward(ctx, mods, xs, groups, feat0, *params):
        ctx.set_materialize_grads(False)
        ep = EncoderPass([m._runner for m in mods], need_feat0=feat0)
        feats, c = ep.forward(list(xs), train=True, groups=list(groups))
        ctx.mods, ctx.ep, ctx.c, ctx.dtype = mods, ep, c, xs[0].dtype
        ctx.nparams = len(params)
        for m in mods:
            m._pending += 1
            if RT.dp is not None:
                RT.dp.note_forward(m)
        # (a lane's features[0] nobody reads is None: the fused stem pass never stored it)
        return tuple(None if f is None else f.permute(0, 3, 1, 2) for lane in feats for f in lane)

    @staticmethod
    def backward(ctx, *g):
        mods = ctx.mods
        nf = len(g) // len(mods)
        gf = [[None if gi is None else nhwc_dense(gi, ctx.dtype) for gi in g[l * nf:(l + 1) * nf]] for l in range(len(mods))]
        ctx.ep.backward(ctx.c, gf)
        ctx.c = None
        for m in mods:
            m._pending -= 1
            if m._pending == 0 and RT.dp is not None:
                RT.dp.grads_ready(m)
        return (None, None, None, None) + (None,) * ctx.nparams


def lanes_compatible(a, b):
    """may two ResNets run as the lanes of one pass (same layers, BatchNorm modes and trained parameters)?"""
    return (isinstance(a, ResNet) and isinstance(b, ResNet) and a.training and b.training
            and a._runner.signature(True) == b._runner.signature(True))


def forward_lanes(depth_net, image, pose_net, pairs, pose_feat0=True):
    """depth_net(image) and pose_net.forward_pairs(pairs) of one training step as ONE encoder pass -> (features of the
    depth encoder, features of the stacked pose pairs), each exactly what the separate call returns.  pose_feat0=False:
    the caller does not read the pose encoder's features[0] (the pose decoder takes the last feature only,
    pose_decoder.py:26-37) — that entry comes back None and its 96 x 320 activation is never stored."""
    require_gpu(image, "ResNet.forward_lanes")
    op_d = depth_net._runner.stem.ready(RT.compute_dtype, image.device)
    op_p = pose_net._runner.stem.ready(RT.compute_dtype, image.device)
    xd = ops.nchw_to_nhwc(image.float(), None, op_d.Ci_p, RT.compute_dtype)
    G, N = len(pairs), image.shape[0]
    xp = torch.empty(G * N, image.shape[2], image.shape[3], op_p.Ci_p, dtype=RT.compute_dtype, device=image.device)
    for g, (a, b) in enumerate(pairs):
        ops.nchw_to_nhwc(a.float(), b.float(), op_p.Ci_p, RT.compute_dtype, out=xp[g * N:(g + 1) * N])
    mods = (depth_net, pose_net)
    for m in mods:
        if m._plist is None:
            m._plist = list(m.parameters())
    outs = _ResNetLanesFn.apply(mods, (xd, xp), (1, G), (True, bool(pose_feat0)), *(depth_net._plist + pose_net._plist))
    nf = len(outs) // 2
    return list(outs[:nf]), list(outs[nf:])


class ResNet(nn.Module):
    planes = [64, 128, 256, 512]

    def __init__(self, block, layers, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(-1, 0, 1, 2, 3), frozen_stages=-1, norm_eval=True, num_input_images=1):
        super().__init__()
        assert 1 <= num_stages <= 4 and max(out_indices) < num_stages
        if tuple(strides) != (1, 2, 2, 2) or any(d != 1 for d in dilations):
            raise NotImplementedError("HIP ResNet supports strides (1,2,2,2) and dilation 1 (the monodepth configs)")
        if tuple(out_indices) != (-1, 0, 1, 2, 3)[: num_stages + 1]:
            raise NotImplementedError("HIP ResNet returns the full pyramid out_indices=(-1,0,1,2,3)")
        self.inplanes = 64
        self.num_stages, self.strides, self.dilations = num_stages, strides, dilations
        self.out_indices, self.frozen_stages = out_indices, frozen_stages
        self.num_input_images, self.norm_eval = num_input_images, norm_eval
        self.conv1 = nn.Conv2d(3 * num_input_images, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i in range(num_stages):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, self.planes[i], layers[i], stride=strides[i]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._runner = ResNetRunner(self)
        self._pending = 0
        self._plist = None
        self.train()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def load_state_dict(self, state_dict, *args, **kwargs):
        # pretrained 3-channel stem tiled over the stacked input images (resnet.py:155-160)
        if 'conv1.weight' in state_dict and self.conv1.weight.shape != state_dict['conv1.weight'].shape:
            state_dict = dict(state_dict)
            state_dict['conv1.weight'] = torch.cat([state_dict['conv1.weight']] * self.num_input_images, 1) / self.num_input_images
        return super().load_state_dict(state_dict, *args, **kwargs)

    def train(self, mode=True):
        super().train(mode)
        if mode:
            self.freeze_stages()
            if self.norm_eval:
                self.freeze_bn()
        return self

    def freeze_stages(self):
        if self.frozen_stages >= 0:
            self.conv1.eval(); self.bn1.eval()
            for p in list(self.conv1.parameters()) + list(self.bn1.parameters()):
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, "layer%d" % i)
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, nn.modules.batchnorm._BatchNorm):
                layer.eval()

    # ---------------------------------------------------------------- execution
    def _run(self, x, groups=1):
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            if self._plist is None:
                self._plist = list(self.parameters())
            return list(_ResNetFn.apply(self, x, groups, *self._plist))
        with torch.no_grad():
            feats, _ = self._runner.forward(x, train=self.bn1.training, groups=groups)
        return [f.permute(0, 3, 1, 2) for f in feats]

    def forward(self, img_batch):
        require_gpu(img_batch, "ResNet.forward")
        op = self._runner.stem.ready(RT.compute_dtype, img_batch.device)
        return self._run(ops.nchw_to_nhwc(img_batch.float(), None, op.Ci_p, RT.compute_dtype))

    def forward_pair(self, a, b):
        """cat([a, b], 1) fused into the layout conversion (pose encoder input, monodepth2_model.py:29-35)."""
        require_gpu(a, "ResNet.forward_pair")
        op = self._runner.stem.ready(RT.compute_dtype, a.device)
        return self._run(ops.nchw_to_nhwc(a.float(), b.float(), op.Ci_p, RT.compute_dtype))


    def forward_pairs(self, pairs):
        """[(a0, b0), (a1, b1), ...] -> features of the stacked batch [G*N, ...]: numerically G separate
        forward_pair calls in list order (per-call BatchNorm batch statistics and running-statistic updates),
        executed as ONE pass — half the launches of the pose encoder and twice the rows per launch."""
        a0 = pairs[0][0]
        require_gpu(a0, "ResNet.forward_pairs")
        op = self._runner.stem.ready(RT.compute_dtype, a0.device)
        G, N = len(pairs), a0.shape[0]
        x = torch.empty(G * N, a0.shape[2], a0.shape[3], op.Ci_p, dtype=RT.compute_dtype, device=a0.device)
        for g, (a, b) in enumerate(pairs):
            ops.nchw_to_nhwc(a.float(), b.float(), op.Ci_p, RT.compute_dtype, out=x[g * N:(g + 1) * N])
        return self._run(x, groups=G)


_DEPTHS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
           101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


def resnet(depth, pretrained=True, **kwargs):
    if depth not in _DEPTHS:
        raise ValueError('Unsupported model depth, must be one of 18, 34, 50, 101, 152')
    block, layers = _DEPTHS[depth]
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        import torch.utils.model_zoo as model_zoo
        urls = {18: 'resnet18-5c106cde', 34: 'resnet34-333f7ec4', 50: 'resnet50-19c8e357',
                101: 'resnet101-5d3b4d8f', 152: 'resnet152-b121ed2d'}
        model.load_state_dict(model_zoo.load_url('https://download.pytorch.org/models/%s.pth' % urls[depth],
                                                 model_dir='.'), strict=False)
    return model


def resnet18(pretrained=True, **kw): return resnet(18, pretrained, **kw)
def resnet34(pretrained=True, **kw): return resnet(34, pretrained, **kw)
def resnet50(pretrained=True, **kw): return resnet(50, pretrained, **kw)
