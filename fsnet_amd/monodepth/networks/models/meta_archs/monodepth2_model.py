"""Meta-archs with the reference's constructor contract (monodepth2_model.py:8-148):
MonoDepthMeta (learned pose, "depth+pose") and MonoDepthWPose (dataset pose).  Sub-networks are built
through build(**cfg) exactly like the reference, so configs only change `name=` strings."""

import torch

from fsnet_amd.engine.runtime import RT, ParamArena, register_arena
from fsnet_amd.vision_base.networks.models.meta_archs.base_meta import BaseMetaArch
from fsnet_amd.vision_base.utils.builder import build


from fsnet_amd.engine import torch_compat as _torch_compat

_torch_compat.install()      # scripts/train.py:100-102 wraps the meta-arch in SyncBatchNorm / DistributedDataParallel


class _HipMetaArch(BaseMetaArch):
    """shared plumbing: flat parameter arena, data-parallel context, compute dtype."""

    def _post_init(self, kwargs):
        if "compute_dtype" in kwargs:
            RT.set_compute_dtype(kwargs["compute_dtype"])
        self._arena = None

    def ensure_arena(self):
        """Flatten all parameters into one fp32 arena (lazily, once they live on the GPU)."""
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("fsnet_amd meta-archs run on MI355X only: call .cuda() before the first forward")
        if self._arena is None or not self._arena.intact():
            self._arena = ParamArena(list(self.named_parameters()), p0.device)
            register_arena(self._arena)
        return self._arena

    def _begin_train(self):
        self.ensure_arena()
        RT.even_chains = False      # (MonoDepthMeta.forward_train: the same encoder architecture on each of two streams)
        if RT.dp is None and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            from fsnet_amd.engine.dataparallel import DataParallelContext
            RT.dp = DataParallelContext(self)
        if RT.dp is not None:
            RT.dp.begin_step(self)
        # re-pack stale MFMA weight operands once per step, on a stream of its own beside the step's weight-free head
        from fsnet_amd.engine.nets import pack_everything_async
        pack_everything_async(self._arena)

    def stage_step_inputs(self, data):
        """non-tensor per-step inputs (fisheye calibrations): host-side staging, also ahead of a hipGraph replay"""
        fn = getattr(self.head, "stage_step_inputs", None)
        if fn is not None:
            fn(data)

    def dummy_forward(self, image):
        if torch.onnx.is_in_onnx_export():
            # scripts/onnx_export.py traces this method: hand the tracer the graph description (export/onnx_graph.py)
            from fsnet_amd.export.onnx_graph import describe_depth_network
            return describe_depth_network(self, image)
        features = self.depth_backbone(image)
        outputs = self.head.forward_depth(features)
        return self.head.get_prediction(None, outputs)


class MonoDepthMeta(_HipMetaArch):
    def __init__(self, depth_backbone_cfg, pose_backbone_cfg, head_cfg, train_cfg, test_cfg, **kwargs):
        super().__init__()
        self.depth_backbone = build(**depth_backbone_cfg)
        self.pose_backbone = build(**pose_backbone_cfg)
        self.head = build(frame_ids=train_cfg.frame_ids, **head_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._post_init(kwargs)

    def _pose_pairs(self, data, image_0):
        fids = list(self.train_cfg.frame_ids[1:])
        return fids, [(data[('image', f_i)], image_0) if f_i < 0 else (image_0, data[('image', f_i)]) for f_i in fids]

    def _lanes_ok(self, image_0):
        """the depth encoder and the stacked pose encoder as the two lanes of ONE pass (engine/nets.py, EncoderPass):
        same architecture, BatchNorm modes and trained parameters, gradients wanted, and a pose head that takes the
        stacked feature"""
        return bool(RT.resolve_lanes()) and self.lanes_possible(image_0)

    def lanes_possible(self, image_0=None):
        """could this model run the two-lane encoder pass at all (what the training hook's autotune asks — before the batch
        is on the device, hence without an image — before it times both arrangements)"""
        if not (RT.batch_pose_pairs and (image_0 is None or image_0.is_cuda) and torch.is_grad_enabled()):
            return False
        if len(self.train_cfg.frame_ids) < 3 or not hasattr(self.head, "forward_pose_pairs"):
            return False
        from fsnet_amd.vision_base.networks.models.backbone.resnet import lanes_compatible
        key = (self.depth_backbone.training, self.pose_backbone.training)
        c = self.__dict__.get("_lanes_cache")
        if c is None or c[0] != key:
            ok = lanes_compatible(self.depth_backbone, self.pose_backbone) and \
                any(p.requires_grad for p in self.depth_backbone.parameters()) and \
                any(p.requires_grad for p in self.pose_backbone.parameters())
            c = self.__dict__["_lanes_cache"] = (key, ok)
        return c[1]

    def _same_encoders(self):
        """both encoders the same trained architecture (RT.even_chains)?"""
        from fsnet_amd.vision_base.networks.models.backbone.resnet import lanes_compatible
        key = (self.depth_backbone.training, self.pose_backbone.training)
        c = self.__dict__.get("_same_enc_cache")
        if c is None or c[0] != key:
            c = self.__dict__["_same_enc_cache"] = (key, bool(lanes_compatible(self.depth_backbone, self.pose_backbone)))
        return c[1]

    def _pose_chain(self, data, image_0, outputs, stacked=None):
        fids, pairs = self._pose_pairs(data, image_0)
        if stacked is None and RT.batch_pose_pairs and len(pairs) > 1 and hasattr(self.pose_backbone, "forward_pairs"):
            # one encoder pass over all pairs, BatchNorm statistics per pair (= separate calls, in this order)
            stacked = self.pose_backbone.forward_pairs(pairs)
        B = image_0.shape[0]
        if stacked is not None and hasattr(self.head, "forward_pose_pairs"):
            res = self.head.forward_pose_pairs([stacked], [f_i < 0 for f_i in fids])
            for f_i, (axisangle, translation, T) in zip(fids, res):
                outputs[("axisangle", f_i)] = axisangle
                outputs[("translation", f_i)] = translation
                outputs[("cam_T_cam", f_i)] = T
            return
        for k, f_i in enumerate(fids):
            pair = pairs[k]
            if stacked is not None:
                pose_feats = [[f[k * B:(k + 1) * B] for f in stacked]]
            elif hasattr(self.pose_backbone, "forward_pair"):
                pose_feats = [self.pose_backbone.forward_pair(*pair)]
            else:
                pose_feats = [self.pose_backbone(torch.cat(pair, 1))]
            axisangle, translation, T = self.head.forward_pose_transform(pose_feats, invert=(f_i < 0))
            outputs[("axisangle", f_i)] = axisangle
            outputs[("translation", f_i)] = translation
            outputs[("cam_T_cam", f_i)] = T

    def _forward_train_lanes(self, data, image_0):
        """One encoder pass for both networks (every post-stem launch serves the depth encoder's 12 images and the pose
        encoder's 24), then the pose decoder on the side stream beside the depth decoder.  The image-only inputs of the
        loss chain (colour pyramid, identity reprojection terms) run on the side stream beside the encoders."""
        from fsnet_amd.vision_base.networks.models.backbone.resnet import forward_lanes
        pose_out = {}
        overlap = RT.overlap
        if overlap:
            main = torch.cuda.current_stream(image_0.device)
            side = RT.side_stream(image_0.device)
            if hasattr(self.head, "prefetch_loss_inputs"):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    RT.mark("side.fork")
                    self.head.prefetch_loss_inputs(data)
        RT.mark("depth.fwd.start")
        _, pairs = self._pose_pairs(data, image_0)
        features, stacked = forward_lanes(self.depth_backbone, image_0, self.pose_backbone, pairs, pose_feat0=False)
        RT.mark("denc.fwd.end")
        if overlap:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                RT.mark("pose.fwd.start")
                self._pose_chain(data, image_0, pose_out, stacked=stacked)
                RT.mark("pose.fwd.end")
        else:
            self._pose_chain(data, image_0, pose_out, stacked=stacked)
        outputs = self.head.forward_depth(features)
        RT.mark("ddec.fwd.end")
        if overlap:
            main.wait_stream(side)            # join before the loss consumes cam_T_cam
            RT.mark("join")
            for v in pose_out.values():
                v.record_stream(main)
        outputs.update(pose_out)
        return self.head.loss(outputs, data)

    def forward_train(self, data, meta):
        self._begin_train()
        image_0 = data[('image', 0)]
        if self._lanes_ok(image_0):
            return self._forward_train_lanes(data, image_0)
        pose_out = {}
        overlap = RT.overlap and image_0.is_cuda
        if overlap:
            # fork: pose chain (encoder pass over both pairs + pose decoder) on the side stream, depth chain on the main
            # one; autograd replays each chain's backward on the stream its forward ran on, so the backward overlaps too.
            # The pose chain is issued first (issued second it starts ~0.2 ms late in the replayed step), and the loss
            # chain's image-only inputs (colour pyramid, identity reprojection terms) follow it on its stream: they need
            # only the batch and are first read by the loss, so they fill the pose stream's idle tail beside the depth
            # decoder instead of delaying both encoders at the head of the step (measured both ways).
            main = torch.cuda.current_stream(image_0.device)
            side = RT.side_stream(image_0.device)
            side.wait_stream(main)
            RT.even_chains = self._same_encoders()
            with torch.cuda.stream(side):
                RT.mark("side.fork")
                RT.mark("pose.fwd.start")
                self._pose_chain(data, image_0, pose_out)
                RT.mark("pose.fwd.end")
                if hasattr(self.head, "prefetch_loss_inputs"):
                    self.head.prefetch_loss_inputs(data)
        RT.mark("depth.fwd.start")
        features = self.depth_backbone(image_0)
        RT.mark("denc.fwd.end")
        outputs = self.head.forward_depth(features)
        RT.mark("ddec.fwd.end")
        if overlap:
            main.wait_stream(side)            # join before the loss consumes cam_T_cam
            RT.mark("join")
            for v in pose_out.values():
                v.record_stream(main)
        else:
            self._pose_chain(data, image_0, pose_out)
        outputs.update(pose_out)
        return self.head.loss(outputs, data)

    def forward_test(self, data, meta):
        features = self.depth_backbone(data[('image', 0)])
        outputs = self.head.forward_depth(features)
        return self.head.get_prediction(data, outputs)


class MonoDepthWPose(_HipMetaArch):
    def __init__(self, depth_backbone_cfg, head_cfg, train_cfg, test_cfg, pose_backbone_cfg=None, **kwargs):
        super().__init__()
        self.depth_backbone = build(**depth_backbone_cfg)
        self.head = build(frame_ids=train_cfg.frame_ids, **head_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.is_use_res_pose = pose_backbone_cfg is not None
        if self.is_use_res_pose:
            raise NotImplementedError("MonoDepthWPose residual-pose branch (pose_backbone_cfg) is not implemented; "
                                      "no shipped config uses it")
        self._post_init(kwargs)

    def forward_train(self, data, meta):
        self._begin_train()
        if list(getattr(self.train_cfg, 'depth_production_frames', [0])) != [0]:
            raise NotImplementedError("depth_production_frames other than [0]")
        features = self.depth_backbone(data[('image', 0)])
        outputs = self.head.forward_depth(features, None if getattr(self.head.depth_decoder, "base_fx", None) is None
                                          else data['P2'])
        for f_i in self.train_cfg.frame_ids[1:]:
            outputs[("cam_T_cam", f_i)] = data[('relative_pose', f_i)]
        return self.head.loss(outputs, data)

    def forward_test(self, data, meta):
        features = self.depth_backbone(data[('image', 0)])
        outputs = self.head.forward_depth(features, None if getattr(self.head.depth_decoder, "base_fx", None) is None
                                          else data['P2'])
        return self.head.get_prediction(data, outputs)


class DistillWPoseMeta(_HipMetaArch):
    """Second training stage (monodepth2_model.py:150-206): a frozen teacher network supplies
    ('teacher_depth', s, s); the student (depth backbone + MultiChannelDepthDecoderUncertain head) is trained with
    dataset poses, the photometric loss and the (uncertainty-weighted) L1 distillation term."""

    def __init__(self, teacher_net_cfg, depth_backbone_cfg, teacher_net_path, head_cfg, train_cfg, test_cfg, **kwargs):
        super().__init__()
        self.teacher_net = build(**teacher_net_cfg)
        if teacher_net_path is not None:
            self.teacher_net.load_state_dict(torch.load(teacher_net_path, map_location='cpu'), strict=False)
        for param in self.teacher_net.parameters():
            param.requires_grad = False
        self.depth_backbone = build(**depth_backbone_cfg)
        self.head = build(frame_ids=train_cfg.frame_ids, **head_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._post_init(kwargs)

    def train(self, mode=True):
        super().train(mode)
        self.teacher_net.eval()
        return self

    def forward_train(self, data, meta):
        self._begin_train()
        image_0 = data[('image', 0)]
        features = self.depth_backbone(image_0)
        outputs = self.head.forward_depth(features, data['P2'])
        outputs.update(self.teacher_net.compute_teacher_depth(image_0))
        for f_i in self.train_cfg.frame_ids[1:]:
            outputs[("cam_T_cam", f_i)] = data[('relative_pose', f_i)]
        return self.head.loss(outputs, data)

    def forward_test(self, data, meta):
        features = self.depth_backbone(data[('image', 0)])
        outputs = self.head.forward_depth(features, data['P2'])
        return self.head.get_prediction(data, outputs)
