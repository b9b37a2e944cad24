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


class ConcatDataset(torch.utils.data.Dataset):
    """Several datasets behind one index range (reference dataset_utils.py:30-56; configs/multi_dataset_example):
    `cfg_list` holds one build() config per child, `common_keywords` are defaults every child config may override.
    Index i belongs to the child whose cumulative start is the last one <= i."""

    def __init__(self, cfg_list, **common_keywords):
        super().__init__()
        from fsnet_amd.vision_base.utils.builder import build
        self.children = [build(**dict(common_keywords, **item)) for item in cfg_list]
        lengths = [len(c) for c in self.children]
        self.seperator = np.concatenate([[0], np.cumsum(lengths[:-1])]).astype(np.int64)   # (the reference's spelling)
        self.total_length = int(sum(lengths))

    def __len__(self):
        return self.total_length

    def _determine_index(self, index):
        child = int(np.searchsorted(self.seperator, index, side="right")) - 1
        return child, index - int(self.seperator[child])

    def __getitem__(self, index):
        child, local = self._determine_index(index)
        return self.children[child][local]
