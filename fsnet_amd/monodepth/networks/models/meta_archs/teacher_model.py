"""MonoDepthInference — the frozen teacher of the self-distillation stage — with the reference's constructor and
methods (monodepth/networks/models/meta_archs/teacher_model.py:5-32), executed by the HIP engine (eval-mode
BatchNorm: DistillWPoseMeta keeps it in .eval())."""
import torch
import torch.nn as nn

from fsnet_amd.vision_base.utils.builder import build


class MonoDepthInference(nn.Module):
    def __init__(self, backbone_cfg, depth_head_cfg, is_produce_detached=True, **kwargs):
        super().__init__()
        self.depth_backbone = build(**backbone_cfg)
        self.depth_decoder = build(**depth_head_cfg)
        self.is_produce_detached = is_produce_detached

    def forward(self, x):
        return self.depth_decoder(self.depth_backbone(x))

    def compute_teacher_depth(self, x):
        """{("teacher_depth", s, s): depth of scale s} — what DistillWPoseMeta feeds compute_distill_loss
        (monodepth2_model.py:177-180).  The frozen teacher runs without an autograd graph unless the config asks otherwise."""
        grad_mode = torch.enable_grad() if not self.is_produce_detached else torch.no_grad()
        with grad_mode:
            predicted = self.forward(x)
        return {("teacher_depth",) + tuple(k[1:3]): v for k, v in predicted.items() if k[0] == "depth"}
