"""build_dataloader (reference vision_base/data/dataloader/dataloader_builder.py:5-17): sampler from the
registry, drop_last=True.  pin_memory is switched on so the hook's H2D copies are asynchronous."""
from torch.utils.data import DataLoader

from ...utils.builder import build

_DEFAULT_SAMPLER = 'fsnet_amd.vision_base.data.dataloader.distributed_sampler.TrainingSampler'


def build_dataloader(dataset, num_workers, batch_size, collate_fn, local_rank=-1, world_size=1, sampler_cfg=None,
                     **kwargs):
    sampler_cfg = dict(sampler_cfg or {})
    name = sampler_cfg.pop('name', _DEFAULT_SAMPLER)
    sampler = build(name, size=len(dataset), rank=local_rank, world_size=world_size, **sampler_cfg)
    kwargs.setdefault("pin_memory", True)
    return DataLoader(dataset, num_workers=num_workers, batch_size=batch_size, collate_fn=collate_fn, sampler=sampler,
                      drop_last=True, **kwargs)
