"""Rank-strided infinite-stream sampler (reference vision_base/data/dataloader/distributed_sampler.py:6-56):
every rank draws the SAME permutation (shared default-seeded generator) and keeps perm[rank::world]."""
import itertools

import torch
from torch.utils.data.sampler import Sampler


class TrainingSampler(Sampler):
    def __init__(self, size, rank=-1, world_size=1, shuffle=True):
        if not isinstance(size, int):
            raise TypeError("TrainingSampler(size=) expects an int. Got type %s." % type(size))
        if size <= 0:
            raise ValueError("TrainingSampler(size=) expects a positive int. Got %s." % size)
        self._size, self._shuffle = size, shuffle
        self._rank, self._world_size = rank, world_size
        self.generator = torch.Generator()

    def __len__(self):
        return self._size

    def __iter__(self):
        order = (torch.randperm(self._size, generator=self.generator) if self._shuffle
                 else torch.arange(self._size)).tolist()
        return itertools.islice(iter(order), max(self._rank, 0), None, self._world_size)
