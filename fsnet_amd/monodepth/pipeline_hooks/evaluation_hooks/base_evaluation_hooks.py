"""KittiEvaluationHook with the reference's constructor and call signature
(monodepth/pipeline_hooks/evaluation_hooks/base_evaluation_hooks.py:19-67): eval-mode forward over the validation set,
crop to the effective size, inverse-depth resize to the original image size, per-image errors, mean + log.
Everything between the network output and the 15 numbers per image stays on the device (fs_resize_linear with
invert, fs_depth_eval); only those numbers are copied back.  The post-optimisation variant (:69-127, sparse visual
odometry) is out of scope."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from fsnet_amd.hip import ops
from fsnet_amd.vision_base.data.datasets.dataset_utils import collate_fn
from fsnet_amd.vision_base.utils.builder import build


class KittiEvaluationHook(object):
    def __init__(self, test_run_hook_cfg, dataset_eval_cfg=None, **kwargs):
        self.test_hook = build(**test_run_hook_cfg)
        self.dataset_eval_func = None if dataset_eval_cfg is None else build(**dataset_eval_cfg)
        for key in kwargs:
            setattr(self, key, kwargs[key])

    @torch.no_grad()
    def __call__(self, meta_arch, dataset_val, writer=None, global_step=0, epoch_num=0):
        meta_arch.eval()
        batch_size = getattr(self, 'batch_size', 1)
        num_workers = getattr(self, 'num_workers', 4)
        dataloader = DataLoader(dataset_val, batch_size, shuffle=False, num_workers=num_workers, collate_fn=collate_fn)
        rows = []
        frame_index = 0
        for batched_data in dataloader:
            output_dict = self.test_hook(batched_data, meta_arch, global_step, epoch_num)
            depth_b = output_dict['depth']
            for i in range(depth_b.shape[0]):
                depth = depth_b[i, 0]
                h_eff, w_eff = (int(v) for v in batched_data[('image_resize', 'effective_size')][i])
                depth = depth[0:h_eff, 0:w_eff].float().contiguous()
                h, w = (int(v) for v in batched_data[('original_image', 0)][i].shape[:2])
                depth_0 = ops.resize_linear(depth, h, w, invert=True)          # 1 / cv2.resize(1 / depth, (w, h))
                gt = self.dataset_eval_func._gt(frame_index, depth_0.device)
                rows.append(ops.depth_eval(depth_0[None], gt[None])[0])
                frame_index += 1
        res = torch.stack(rows).cpu().numpy()
        if (res[:, 15] == 0).any():
            raise ValueError
        mean_errors, mean_abs_errors = res[:, 1:8].mean(0), res[:, 8:15].mean(0)
        self.dataset_eval_func.log(writer, mean_errors, mean_abs_errors, global_step=global_step, epoch_num=epoch_num)
        meta_arch.train()
        return dict(mean_errors=mean_errors, mean_abs_errors=mean_abs_errors, ratios=res[:, 0])
