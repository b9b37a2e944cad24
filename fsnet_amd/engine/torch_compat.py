"""Pass-throughs that let the reference's scripts/train.py run UNMODIFIED on fsnet_amd meta-archs.

train.py hard-wires torch's data-parallel wrappers (scripts/train.py:100-102):

    meta_arch = torch.nn.SyncBatchNorm.convert_sync_batchnorm(meta_arch)
    meta_arch = torch.nn.parallel.DistributedDataParallel(meta_arch.cuda(), device_ids=[gpu], output_device=gpu)

An fsnet_amd meta-arch exchanges its BatchNorm statistics and gradient buckets itself (engine/dataparallel.py: RCCL
collectives between its own kernels, written straight into the flat gradient arena), so torch's two wrappers have
nothing to do — worse, DistributedDataParallel's reducer waits for autograd hooks on parameters whose gradients never
pass through autograd here.  Importing the meta-arch module therefore installs two guards, active ONLY for
fsnet_amd meta-archs and transparent for every other module:

  * SyncBatchNorm.convert_sync_batchnorm(module)  -> returns an fsnet_amd meta-arch unchanged
  * DistributedDataParallel(module, ...)          -> for an fsnet_amd meta-arch, a thin wrapper with the same
    `.module` attribute / `module.`-prefixed state_dict / forward signature and no reducer

plus `adopt_optimizer`: the reference's build_optimizer returns torch.optim.Adam (optimizers.py:7-8); the training
hook adopts such an instance into the fused clip+Adam kernel, sharing its param_groups (schedulers keep working) and
its state dict (checkpoints keep torch.optim.Adam's format).
"""
import torch
import torch.nn as nn

_installed = False


class HipDataParallel(nn.Module):
    """what DistributedDataParallel(meta_arch) returns for an fsnet_amd meta-arch: the attribute surface train.py and
    save_models / load_models use (`.module`, forward(data, meta), train(), state_dict with the `module.` prefix)"""

    def __init__(self, module, device_ids=None, output_device=None, **kwargs):
        super().__init__()
        self.module = module
        self.device_ids, self.output_device = device_ids, output_device

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def _is_ours(module):
    from fsnet_amd.monodepth.networks.models.meta_archs.monodepth2_model import _HipMetaArch
    return isinstance(module, _HipMetaArch)


def install():
    global _installed
    if _installed:
        return
    _installed = True
    real_convert = nn.SyncBatchNorm.convert_sync_batchnorm.__func__
    real_ddp = nn.parallel.DistributedDataParallel

    def convert_sync_batchnorm(cls, module, process_group=None):
        if _is_ours(module):
            return module           # SyncBN statistics are exchanged inside the engine (global-batch statistics)
        return real_convert(cls, module, process_group)

    class DistributedDataParallel(real_ddp):
        def __new__(cls, module=None, *args, **kwargs):
            if module is not None and _is_ours(module):
                return HipDataParallel(module, *args, **kwargs)
            return super().__new__(cls)

    DistributedDataParallel.__name__ = real_ddp.__name__
    DistributedDataParallel.__qualname__ = real_ddp.__qualname__
    nn.SyncBatchNorm.convert_sync_batchnorm = classmethod(convert_sync_batchnorm)
    nn.parallel.DistributedDataParallel = DistributedDataParallel
    import torch.nn.parallel.distributed as _d
    _d.DistributedDataParallel = DistributedDataParallel


def adopt_optimizer(optimizer, meta_arch):
    """torch.optim.Adam over exactly the meta-arch's parameters -> FusedAdam sharing its param_groups and state;
    anything else is returned unchanged (the hook then runs it as a plain torch optimizer)."""
    from fsnet_amd.vision_base.networks.optimizers.optimizers import FusedAdam
    if isinstance(optimizer, FusedAdam) or type(optimizer) is not torch.optim.Adam:
        return optimizer
    fused = getattr(optimizer, "_fsnet_fused", None)
    if fused is not None:
        return fused
    if len(optimizer.param_groups) != 1:
        return optimizer
    g = optimizer.param_groups[0]
    if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
        return optimizer
    inner = getattr(meta_arch, "module", meta_arch)
    if [id(p) for p in g["params"]] != [id(p) for p in inner.parameters()]:
        return optimizer
    fused = FusedAdam(g["params"], lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], model=inner)
    fused.param_groups = optimizer.param_groups      # one list: a scheduler stepping the torch optimizer moves both
    fused.state = optimizer.state                    # one dict: checkpoints written from either see the moments
    optimizer.state_dict = fused.state_dict          # (refreshes the per-parameter step counts first)
    real_load = optimizer.load_state_dict

    def load_state_dict(sd):
        real_load(sd)
        fused.state, fused.param_groups = optimizer.state, optimizer.param_groups
        fused._readopt = True
    optimizer.load_state_dict = load_state_dict
    optimizer._fsnet_fused = fused
    return fused
