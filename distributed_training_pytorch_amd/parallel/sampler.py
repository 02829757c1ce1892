"""Deterministic per-rank index sharding (the DistributedSampler role).

Parity target: reference ``trainer/trainer.py:215`` uses ``DistributedSampler(dataset,
shuffle=True)`` and ``:140`` calls ``sampler.set_epoch(epoch)``.  This sampler produces the *same
index sequence* as ``torch.utils.data.DistributedSampler`` with seed 0 (permutation from
``torch.Generator().manual_seed(seed + epoch)``, padded by wrapping to a multiple of the world
size, strided slice ``rank::world``) — tests/test_unit_cpu.py pins that — but takes rank/world
explicitly, so it works without a process group (single-GPU runs, unit tests), and supports
``drop_last`` and unpadded shards (``pad=False``: validation without duplicated samples).
"""
import math

import torch
from torch.utils.data import Sampler


class ShardedSampler(Sampler):
    def __init__(self, dataset_len, rank=0, world_size=1, shuffle=True, seed=0, drop_last=False, pad=True):
        if hasattr(dataset_len, "__len__"):
            dataset_len = len(dataset_len)
        if world_size < 1 or not (0 <= rank < world_size):
            raise ValueError("invalid rank %r for world size %r" % (rank, world_size))
        self.n = int(dataset_len)
        self.rank = int(rank)
        self.world_size = int(world_size)
        self.shuffle = bool(shuffle)
        self.seed = int(seed)
        self.drop_last = bool(drop_last)
        self.pad = bool(pad)        # pad=False: no duplicated samples; shards may differ by one
        self.epoch = 0
        if self.drop_last and self.n % self.world_size != 0:
            self.num_samples = math.ceil((self.n - self.world_size) / self.world_size)
        else:
            self.num_samples = math.ceil(self.n / self.world_size)
        self.total_size = self.num_samples * self.world_size
        if not self.pad and not self.drop_last:
            self.total_size = self.n
            self.num_samples = len(range(self.rank, self.n, self.world_size))

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def global_order(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.pad and not self.drop_last:
            return idx
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad > 0:
                if pad <= len(idx):
                    idx += idx[:pad]
                else:
                    idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
        else:
            idx = idx[:self.total_size]
        return idx

    def __iter__(self):
        return iter(self.global_order()[self.rank:self.total_size:self.world_size])

    def __len__(self):
        return self.num_samples
