"""DistributedSampler with torch's exact index semantics (torch/utils/data/distributed.py:98-141), as used by the
reference for train/valid/test (/root/reference/dataloader.py:147-152: shuffle=True, seed 0, drop_last=False).
The permutation comes from torch.Generator/randperm (bit parity with the reference); padding and the rank::world
selection run in the C++ host runtime (csrc/host/sampler_idx_planner.cpp)."""
from __future__ import annotations

import ctypes
import math

import torch

from ..ops import native


class DistributedSampler:
    def __init__(self, dataset_len, num_replicas=1, rank=0, shuffle=True, seed=0):
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.n, self.num_replicas, self.rank, self.shuffle, self.seed = int(dataset_len), num_replicas, rank, shuffle, seed
        self.epoch = 0
        self.num_samples = math.ceil(self.n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self) -> torch.Tensor:
        """int64 CPU tensor of this rank's indices for the current epoch."""
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(self.n, generator=g)
        else:
            perm = torch.arange(self.n)
        out = torch.empty(self.num_samples, dtype=torch.int64)
        if native.have_host_lib():
            k = native.host_lib().dpa_sampler_select(
                ctypes.c_void_p(perm.data_ptr()), ctypes.c_longlong(self.n), self.num_replicas, self.rank,
                ctypes.c_void_p(out.data_ptr()))
            assert k == self.num_samples
        else:                                   # pure-python path (library not built yet)
            pos = torch.arange(self.rank, self.total_size, self.num_replicas) % max(self.n, 1)
            out = perm[pos]
        return out

    def __iter__(self):
        return iter(self.indices().tolist())

    def __len__(self):
        return self.num_samples
