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
        features = self.depth_backbone(x)
        return self.depth_decoder(features)

    def compute_teacher_depth(self, x):
        if self.is_produce_detached:
            with torch.no_grad():
                output_dict = self(x)
        else:
            output_dict = self(x)
        teacher_output = {}
        for key in output_dict:
            if key[0] == 'depth':
                teacher_output[("teacher_depth", key[1], key[2])] = output_dict[key]
        return teacher_output
