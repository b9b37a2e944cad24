"""collate_fn (reference vision_base/data/datasets/dataset_utils.py:7-27): stack tensors / ndarrays of
the keys every sample shares, list everything else."""
import numpy as np
import torch


def find_shared_keys(batch):
    shared = set(batch[0].keys())
    for item in batch[1:]:
        shared &= set(item.keys())
    return list(shared)


def collate_fn(batch):
    out = {}
    for key in find_shared_keys(batch):
        first = batch[0][key]
        if isinstance(first, torch.Tensor):
            out[key] = torch.stack([item[key] for item in batch], dim=0)
        elif isinstance(first, np.ndarray):
            out[key] = torch.stack([torch.from_numpy(item[key]) for item in batch], dim=0)
        else:
            out[key] = [item[key] for item in batch]
    return out
