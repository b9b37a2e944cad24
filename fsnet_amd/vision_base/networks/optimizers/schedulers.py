"""build_scheduler with the reference's names (vision_base/networks/optimizers/schedulers.py:3-94).
Schedulers are host-side scalar bookkeeping over optimizer.param_groups['lr']."""
import torch.optim as optim


class PolyLR(optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, gamma=0.9, n_iteration=-1):
        self.step_size, self.gamma = n_iteration, gamma
        super().__init__(optimizer)

    def get_lr(self):
        decay = max(0., 1 - self._step_count / float(self.step_size)) ** self.gamma
        return [lr * decay for lr in self.base_lrs]


class GradualWarmupScheduler(optim.lr_scheduler._LRScheduler):
    """linear warm-up to base_lr * multiplier over total_epoch, then hands over to after_scheduler."""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler_cfg=None):
        if multiplier < 1.:
            raise ValueError('multiplier should be greater thant or equal to 1.')
        self.multiplier, self.total_epoch = multiplier, total_epoch
        self.after_scheduler = build_scheduler(optimizer, **(after_scheduler_cfg or {}))
        self.finished = False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if not self.finished:
                self.after_scheduler.base_lrs = [lr * self.multiplier for lr in self.base_lrs]
                self.finished = True
            return self.after_scheduler.get_last_lr()
        if self.multiplier == 1.0:
            return [lr * (float(self.last_epoch) / self.total_epoch) for lr in self.base_lrs]
        return [lr * ((self.multiplier - 1.) * self.last_epoch / self.total_epoch + 1.) for lr in self.base_lrs]

    def step(self, epoch=None, metrics=None):
        if self.finished:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
            self._last_lr = self.after_scheduler.get_last_lr()
        else:
            return super().step(epoch)


_TORCH = {"steplr": optim.lr_scheduler.StepLR, "multisteplr": optim.lr_scheduler.MultiStepLR,
          "exponentiallr": optim.lr_scheduler.ExponentialLR, "cosineannealinglr": optim.lr_scheduler.CosineAnnealingLR}


def build_scheduler(optimizer, name=None, **kwargs):
    kwargs.pop("is_iter_based", None)
    if name is None:
        return optim.lr_scheduler.ExponentialLR(optimizer, 1.0)
    key = name.lower()
    if key in _TORCH:
        return _TORCH[key](optimizer, **kwargs)
    if key == "polylr":
        return PolyLR(optimizer, **kwargs)
    if key == "gradualwarmupscheduler":
        return GradualWarmupScheduler(optimizer, **kwargs)
    raise NotImplementedError(name)
