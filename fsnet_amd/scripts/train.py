#!/usr/bin/env python
"""Training driver: the loop of the reference's scripts/train.py:21-214 (config -> build(**cfg...) ->
epochs x iterations of training_hook -> scheduler -> checkpoints -> barrier), on the HIP engine.

    python -m fsnet_amd.scripts.train --config my_cfg.py [--a.b.c=value ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fsnet_amd.scripts.train --config ...

Differences from the reference, all deliberate: no torch SyncBatchNorm/DDP wrappers (the engine exchanges BN
statistics and gradients itself over RCCL), no tensorboard/git requirements (writer optional), evaluation
hooks are out of scope for this path and are skipped unless importable.
"""
import argparse
import ast
import os
import sys

import torch

from fsnet_amd.vision_base.data.dataloader import build_dataloader
from fsnet_amd.vision_base.data.datasets.dataset_utils import collate_fn
from fsnet_amd.vision_base.networks.models.meta_archs.base_meta import BaseMetaArch
from fsnet_amd.vision_base.networks.optimizers import optimizers, schedulers
from fsnet_amd.vision_base.networks.utils.utils import load_models, save_models
from fsnet_amd.vision_base.pipeline_hooks.train_val_hooks.base_training_hooks import BaseTrainingHook
from fsnet_amd.vision_base.utils.builder import build
from fsnet_amd.vision_base.utils.logger import LossLogger
from fsnet_amd.vision_base.utils.timer import Timer
from fsnet_amd.vision_base.utils.utils import cfg_from_file, get_num_parameters, set_random_seed, update_cfg


def parse_overrides(argv):
    out = {}
    for a in argv:
        if a.startswith("--") and "=" in a:
            k, v = a[2:].split("=", 1)
            try:
                v = ast.literal_eval(v)     # literals only, like python-fire in the reference
            except (ValueError, SyntaxError):
                pass
            out[k] = v
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--experiment_name", default="default")
    args, rest = ap.parse_known_args(argv)
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "-1")) if world_size > 1 else -1
    rank = int(os.environ.get("RANK", "0")) if world_size > 1 else -1     # global rank: sampler shard + logging gate
    is_distributed = world_size > 1
    is_logging = rank <= 0

    cfg = update_cfg(cfg_from_file(args.config), **parse_overrides(rest))
    gpu = local_rank if is_distributed else min(getattr(cfg.trainer, "gpu", 0), torch.cuda.device_count() - 1)
    set_random_seed(123)
    torch.cuda.set_device(gpu)
    if is_distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend="nccl", init_method="env://")

    dataset_train = build(**cfg.train_dataset)
    dataloader_train = build_dataloader(dataset_train, num_workers=cfg.data.num_workers, batch_size=cfg.data.batch_size,
                                        collate_fn=collate_fn, local_rank=rank, world_size=world_size,
                                        sampler_cfg=getattr(cfg.data, "sampler", dict()))
    meta_arch = build(**cfg.meta_arch)
    assert isinstance(meta_arch, BaseMetaArch)
    meta_arch = meta_arch.cuda().train()
    if is_logging:
        print("number of trained parameters of the model: %d" % get_num_parameters(meta_arch))

    optimizer = optimizers.build_optimizer(meta_arch, **cfg.optimizer)
    scheduler_config = dict(getattr(cfg, "scheduler", None) or {})
    is_iter_based = scheduler_config.pop("is_iter_based", False)
    scheduler = schedulers.build_scheduler(optimizer, **scheduler_config)
    training_loss_logger = LossLogger(None, "train") if is_logging else None

    old_checkpoint = getattr(cfg.path, "pretrained_checkpoint", None) if "path" in cfg else None
    if old_checkpoint is not None:
        load_models(old_checkpoint, meta_arch, optimizer, map_location="cuda:%d" % gpu)

    if "training_hook" not in cfg.trainer:
        raise KeyError("cfg.trainer.training_hook")
    training_hook = build(**cfg.trainer.training_hook)
    assert isinstance(training_hook, BaseTrainingHook)

    timer = Timer()
    ckpt_dir = getattr(cfg.path, "checkpoint_path", None) if "path" in cfg else None
    max_iters = getattr(cfg.trainer, "max_iters", None)
    try:
        _loop(cfg, meta_arch, optimizer, scheduler, is_iter_based, training_hook, training_loss_logger, dataloader_train,
              timer, ckpt_dir, is_logging, is_distributed, world_size, max_iters)
    finally:
        from fsnet_amd.engine.runtime import RT
        if RT.dp is not None:
            RT.dp.close()                     # the engine's RCCL communicator, before the process group goes
            RT.dp = None
        if is_distributed and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    return meta_arch


def _loop(cfg, meta_arch, optimizer, scheduler, is_iter_based, training_hook, training_loss_logger, dataloader_train,
          timer, ckpt_dir, is_logging, is_distributed, world_size, max_iters):
    global_step = 0
    for epoch_num in range(cfg.trainer.max_epochs):
        meta_arch.train()
        if training_loss_logger:
            training_loss_logger.reset()
        for iter_num, data in enumerate(dataloader_train):
            training_hook(data, meta_arch, optimizer, None, training_loss_logger, global_step, epoch_num)
            global_step += 1
            if is_iter_based:
                scheduler.step()
            if is_logging and global_step % cfg.trainer.disp_iter == 0 and "total_loss" in training_loss_logger.loss_stats:
                print("Epoch: {} | Iteration: {}  | Running loss: {:1.5f} | eta:{}".format(
                    epoch_num, iter_num, training_loss_logger.loss_stats["total_loss"].avg,
                    timer.compute_eta(global_step, len(dataloader_train) * cfg.trainer.max_epochs / world_size)), end="\r")
            if max_iters and global_step >= max_iters:
                break
        if not is_iter_based:
            scheduler.step()
        if is_logging and ckpt_dir:
            os.makedirs(ckpt_dir, exist_ok=True)
            save_models(os.path.join(ckpt_dir, "%s_latest.pth" % cfg.meta_arch.name), meta_arch, optimizer)
            if (epoch_num + 1) % cfg.trainer.save_iter == 0:
                save_models(os.path.join(ckpt_dir, "%s_%d.pth" % (cfg.meta_arch.name, epoch_num)), meta_arch, optimizer)
        if is_distributed:
            torch.distributed.barrier()
        if max_iters and global_step >= max_iters:
            break                              # leave the epoch loop too
    if is_logging:
        print("\nfinished %d steps" % global_step)


if __name__ == "__main__":
    main(sys.argv[1:])
