"""Samplers with the reference's semantics (mono/datasets/loader/sampler.py): `DistributedGroupSampler` (:82-163, the one
`build_dataloader(dist=True, shuffle=True)` uses), `DistributedSampler` (:15-38) and `GroupSampler` (:41-79).
Pure host logic on torch / numpy generators: given the same epoch they emit exactly the reference's index sequence, so
a run here visits the data in the same order as the reference would (tests/golden/sampler.npz holds sequences produced
by the reference classes themselves).  Each rank of the data-parallel job takes a contiguous block of whole per-GPU
batches of an epoch-seeded permutation; groups (`dataset.flag`) never mix inside a batch."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.utils.data import Sampler


def _dist_info(num_replicas, rank):
    import torch.distributed as dist
    if num_replicas is None:
        num_replicas = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if rank is None:
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return num_replicas, rank


class DistributedSampler(Sampler):
    """sampler.py:15-38: epoch-seeded permutation (or arange), padded to a multiple of the world size, strided by rank."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        self.dataset = dataset
        self.num_replicas, self.rank = _dist_info(num_replicas, rank)
        self.epoch, self.shuffle = 0, shuffle
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = torch.arange(len(self.dataset)).tolist()
        indices += indices[:(self.total_size - len(indices))]
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class GroupSampler(Sampler):
    """sampler.py:41-79 (numpy global RNG, like the reference)."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, "flag")
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = dataset.flag.astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(s / samples_per_gpu)) * samples_per_gpu for s in self.group_sizes)

    def __iter__(self):
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            indice = np.where(self.flag == i)[0]
            np.random.shuffle(indice)
            num_extra = int(np.ceil(size / self.samples_per_gpu)) * self.samples_per_gpu - len(indice)
            indices.append(np.concatenate([indice, indice[:num_extra]]))
        indices = np.concatenate(indices)
        indices = np.concatenate([indices[i * self.samples_per_gpu:(i + 1) * self.samples_per_gpu]
                                  for i in np.random.permutation(range(len(indices) // self.samples_per_gpu))])
        assert len(indices) == self.num_samples
        return iter(torch.from_numpy(indices).long())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(Sampler):
    """sampler.py:82-163."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        self.num_replicas, self.rank = _dist_info(num_replicas, rank)
        self.dataset, self.samples_per_gpu, self.epoch = dataset, samples_per_gpu, 0
        assert hasattr(dataset, "flag")
        self.flag = dataset.flag
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = 0
        for size in self.group_sizes:
            self.num_samples += int(math.ceil(size * 1.0 / samples_per_gpu / self.num_replicas)) * samples_per_gpu
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)                 # deterministic shuffle per epoch, identical on every rank
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                indice = np.where(self.flag == i)[0]
                indice = indice[list(torch.randperm(int(size), generator=g))].tolist()
                extra = int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu * \
                    self.num_replicas - len(indice)
                indice += indice[:extra]
                indices += indice
        assert len(indices) == self.total_size
        spg = self.samples_per_gpu
        indices = [indices[j] for i in list(torch.randperm(len(indices) // spg, generator=g))
                   for j in range(i * spg, (i + 1) * spg)]
        offset = self.num_samples * self.rank       # this rank's contiguous block of whole batches
        indices = indices[offset:offset + self.num_samples]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
