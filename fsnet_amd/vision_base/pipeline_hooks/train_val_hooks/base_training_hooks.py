"""One optimisation step, with the reference hook's constructor and call signature
(vision_base/pipeline_hooks/train_val_hooks/base_training_hooks.py:9-49): zero_grad -> H2D -> forward ->
loss.backward() -> clip_grad_norm_ -> optimizer.step().  With the HIP FusedAdam the zero/clip/step
collapse to one memset + two kernels over the flat arena, with no host synchronisation in the step."""
import torch

from fsnet_amd.engine.runtime import RT
from fsnet_amd.vision_base.utils.timer import profile


class BaseTrainingHook(object):
    def __init__(self, tensor_keys=None, clip_gradients=None, **kwargs):
        self.tensor_keys = tensor_keys
        self.clip_gradients = clip_gradients

    @profile('Training hook', 0, 100)
    def __call__(self, data, meta_arch, optimizer, writer=None, training_loss_logger=None, global_step=0,
                 epoch_num=0):
        from fsnet_amd.vision_base.networks.optimizers.optimizers import FusedAdam
        inner = getattr(meta_arch, "module", meta_arch)
        arena = inner.ensure_arena() if hasattr(inner, "ensure_arena") else None
        fused = isinstance(optimizer, FusedAdam)
        if arena is not None:
            arena.zero_grads()
        else:
            optimizer.zero_grad()

        for key in data:
            if isinstance(data[key], torch.Tensor):
                if self.tensor_keys is None or key in self.tensor_keys:
                    data[key] = data[key].cuda(non_blocking=True).contiguous()

        meta = dict(epoch_num=epoch_num, global_step=global_step, is_training=True)
        output = meta_arch(data, meta)

        if training_loss_logger is not None:
            training_loss_logger.update(output['loss_dict'])
            training_loss_logger.update_hm(output.get('hm', dict()))

        loss = output['loss']
        (loss if loss.dim() == 0 else loss.mean()).backward()

        grad_scale = RT.dp.finish() if RT.dp is not None else 1.0
        if fused:
            optimizer.step(max_norm=self.clip_gradients, grad_scale=grad_scale)
        else:
            if grad_scale != 1.0:
                for p in meta_arch.parameters():
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
            if self.clip_gradients is not None:
                torch.nn.utils.clip_grad_norm_(meta_arch.parameters(), self.clip_gradients)
            optimizer.step()
            RT.bump_weights()
        return output
