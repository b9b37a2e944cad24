"""Config dictionaries in FSNet's own format (configs/kitti_wpose_example:170-215; depth+pose wiring
from tests/example_cfgs/config.py:131-180 of the reference) with the `name=` strings repointed to the
fsnet_amd plugin paths — the only change a reference config needs."""
import numpy as np

from .vision_base.utils.utils import EasyDict

P = "fsnet_amd."
NUM_CH_ENC = {18: [64, 64, 128, 256, 512], 34: [64, 64, 128, 256, 512], 50: [64, 256, 512, 1024, 2048]}


def meta_arch_cfg(height=192, width=640, with_pose=True, depth=18, num_output_channels=16, min_depth=0.5,
                  max_depth=100.0, frame_ids=(0, 1, -1), scales=(0, 1, 2, 3), base_fx=None, fisheye=False):
    """with_pose: MonoDepthMeta (learned pose) else MonoDepthWPose (dataset pose).  base_fx: focal-length depth scaling
    (configs/multi_dataset_example:256).  fisheye: FishEyeDecoder head (configs/kitti360_fisheye_example:198-215)."""
    enc = np.array(NUM_CH_ENC[depth])
    backbone = dict(name=P + 'vision_base.networks.models.backbone.resnet.resnet', depth=depth, pretrained=False,
                    frozen_stages=-1, num_stages=4, out_indices=(-1, 0, 1, 2, 3), norm_eval=False,
                    dilations=(1, 1, 1, 1))
    head = dict(
        name=P + 'monodepth.networks.models.heads.monodepth2_decoder.' + ('FishEyeDecoder' if fisheye else 'MonoDepth2Decoder'),
        scales=list(scales),
        height=height, width=width, min_depth=min_depth, max_depth=max_depth, overlapped_mask=True, is_log_image=False,
        depth_decoder_cfg=dict(name=P + 'monodepth.networks.models.heads.depth_encoder.MultiChannelDepthDecoder',
                               num_ch_enc=enc, num_output_channels=num_output_channels, use_skips=True,
                               scales=list(scales), min_depth=min_depth, max_depth=max_depth))
    if base_fx is not None:
        head['depth_decoder_cfg']['base_fx'] = base_fx
    cfg = dict(depth_backbone_cfg=backbone, head_cfg=head, train_cfg=EasyDict(frame_ids=list(frame_ids)),
               test_cfg=EasyDict())
    if with_pose:
        head['pose_decoder_cfg'] = dict(name=P + 'monodepth.networks.models.heads.pose_decoder.PoseDecoder',
                                        num_ch_enc=enc, num_input_features=1, num_frames_to_predict_for=2, stride=1)
        cfg['pose_backbone_cfg'] = dict(backbone, num_input_images=2)
        cfg['name'] = P + 'monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthMeta'
    else:
        cfg['name'] = P + 'monodepth.networks.models.meta_archs.monodepth2_model.MonoDepthWPose'
    return EasyDict(cfg)


def training_cfg(clip_gradients=35.0, lr=1e-4, weight_decay=0):
    return EasyDict(
        training_hook=dict(name=P + 'vision_base.pipeline_hooks.train_val_hooks.base_training_hooks.BaseTrainingHook',
                           clip_gradients=clip_gradients),
        optimizer=dict(name='adam', lr=lr, weight_decay=weight_decay),
        scheduler=dict(name='StepLR', step_size=15))
