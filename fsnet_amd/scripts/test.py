"""Evaluation driver (mirror of the reference's scripts/test.py:12-55): config -> dataset split -> meta-arch ->
checkpoint (strict=False) -> cfg.trainer.evaluate_hook(meta_arch, dataset).

    python -m fsnet_amd.scripts.test --config CFG --checkpoint_path CKPT [--gpu 0] [--split_to_test validation]
                                     [--a.b.c=value ...]      (config overrides, as in scripts/train.py)
"""
import argparse
import ast

import torch

from fsnet_amd.vision_base.networks.utils.utils import load_models
from fsnet_amd.vision_base.utils.builder import build
from fsnet_amd.vision_base.utils.utils import cfg_from_file, update_cfg

_SPLITS = {"training": "train_dataset", "test": "test_dataset"}      # anything else: the validation split


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config/config.py")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--checkpoint_path", default="retinanet_79.pth")
    ap.add_argument("--split_to_test", default="validation")
    args, extra = ap.parse_known_args(argv)
    overrides = {}
    for item in extra:
        key, _, v = item.lstrip("-").partition("=")
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        overrides[key] = v
    cfg = update_cfg(cfg_from_file(args.config), **overrides)
    cfg.trainer.gpu = args.gpu
    torch.cuda.set_device(cfg.trainer.gpu)
    dataset = build(**cfg[_SPLITS.get(args.split_to_test, "val_dataset")])
    meta_arch = build(**cfg.meta_arch).cuda()
    load_models(args.checkpoint_path, meta_arch, map_location="cuda:%d" % args.gpu, strict=False)
    meta_arch.eval()
    if "evaluate_hook" not in cfg.trainer:
        raise KeyError("evaluate_hook not found in Config")
    evaluate_hook = build(result_path_split="validation", **cfg.trainer.evaluate_hook)
    evaluate_hook(meta_arch, dataset)
    print("finish")
    return meta_arch


if __name__ == "__main__":
    main()
