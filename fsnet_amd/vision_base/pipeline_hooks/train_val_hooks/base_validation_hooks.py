"""BaseValidationHook with the reference's constructor and call signature
(vision_base/pipeline_hooks/train_val_hooks/base_validation_hooks.py:5-30): move the batch to the device, run the
meta-arch with is_training=False (eval-mode BatchNorm through the HIP engine), return its output dict."""
import torch


class BaseValidationHook(object):
    def __init__(self, tensor_keys=None, **kwargs):
        self.tensor_keys = tensor_keys

    def __call__(self, data, meta_arch, global_step=0, epoch_num=0):
        for key in data:
            if isinstance(data[key], torch.Tensor):
                if self.tensor_keys is None or key in self.tensor_keys:
                    data[key] = data[key].cuda(non_blocking=True).contiguous()
        meta = dict(epoch_num=epoch_num, global_step=global_step, is_training=False)
        with torch.no_grad():
            return meta_arch(data, meta)
