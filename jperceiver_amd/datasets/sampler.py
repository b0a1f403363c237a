"""Index samplers that visit the data in the reference's order (mono/datasets/loader/sampler.py: `DistributedSampler`
:15-38, `GroupSampler` :41-79, `DistributedGroupSampler` :82-163 -- the one `build_dataloader(dist=True)` uses).

Own structure: every sampler is "a plan of whole per-GPU batches" built from three array helpers --

    _groups(flag)            members of each non-empty group (`dataset.flag`), ascending group id, as index arrays
    _cycle_to(idx, n)        idx extended cyclically with its own head to length n (the reference's padding)
    _reorder_batches(...)    a flat plan cut into batches of `b` and emitted in a given batch order

-- and the only thing taken over from the reference is what index-exact parity forces: which random generator is asked
for which permutation, in which order (tests/golden/sampler.npz holds 78 sequences emitted by the reference's own
classes).  Ranks of a data-parallel job own a contiguous run of whole batches of the epoch plan; a batch never mixes
groups."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Sampler


def _groups(flag) -> list:
    flag = np.asarray(flag, dtype=np.int64)
    by_group = np.argsort(flag, kind="stable")                       # members of group 0, then 1, ... each ascending
    counts = np.bincount(flag)
    return [m for m in np.split(by_group, np.cumsum(counts)[:-1]) if len(m)]


def _cycle_to(idx: np.ndarray, n: int) -> np.ndarray:
    return np.resize(idx, n) if n != len(idx) else idx


def _ceil_to(n: int, multiple: int) -> int:
    return -(-n // multiple) * multiple


def _reorder_batches(plan: np.ndarray, b: int, batch_order) -> np.ndarray:
    return plan.reshape(-1, b)[np.asarray(batch_order, dtype=np.int64)].reshape(-1)


def _world(num_replicas, rank):
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    return (num_replicas if num_replicas is not None else (dist.get_world_size() if live else 1),
            rank if rank is not None else (dist.get_rank() if live else 0))


class _EpochSampler(Sampler):
    """Common shell: a per-epoch plan (`_plan() -> int64 array`) of `num_samples` indices, `set_epoch`."""
    epoch = 0
    num_samples = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        plan = self._plan()
        if len(plan) != self.num_samples:
            raise RuntimeError(f"{type(self).__name__}: planned {len(plan)} indices, expected {self.num_samples}")
        return iter(plan.tolist())

    def _epoch_generator(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)          # identical on every rank: the ranks cut ONE shared plan
        return g


class DistributedSampler(_EpochSampler):
    """Whole data set, epoch-seeded permutation (or identity), padded cyclically to a multiple of the world size, rank r
    takes elements r, r + world, ..."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        self.dataset, self.shuffle = dataset, shuffle
        self.num_replicas, self.rank = _world(num_replicas, rank)
        self.total_size = _ceil_to(len(dataset), self.num_replicas)
        self.num_samples = self.total_size // self.num_replicas

    def _plan(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._epoch_generator()).numpy() if self.shuffle else np.arange(n)
        return _cycle_to(order, self.total_size)[self.rank::self.num_replicas]


class GroupSampler(_EpochSampler):
    """Single process: each group shuffled (numpy's global generator, as the reference), padded to whole batches; then the
    batches of all groups are shuffled among each other."""

    def __init__(self, dataset, samples_per_gpu=1):
        if not hasattr(dataset, "flag"):
            raise AttributeError("GroupSampler needs dataset.flag")
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = int(sum(_ceil_to(int(s), samples_per_gpu) for s in self.group_sizes))

    def _plan(self):
        b = self.samples_per_gpu
        parts = []
        for members in _groups(self.flag):
            members = members.copy()
            np.random.shuffle(members)
            parts.append(_cycle_to(members, _ceil_to(len(members), b)))
        plan = np.concatenate(parts)
        return _reorder_batches(plan, b, np.random.permutation(len(plan) // b))


class DistributedGroupSampler(_EpochSampler):
    """Data parallel: each group permuted with the epoch generator and padded to whole batches for EVERY rank, all batches
    shuffled with the same generator, rank r owns batches [r * k, (r + 1) * k)."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        if not hasattr(dataset, "flag"):
            raise AttributeError("DistributedGroupSampler needs dataset.flag")
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.num_replicas, self.rank = _world(num_replicas, rank)
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag)
        per_round = samples_per_gpu * self.num_replicas              # one batch on every rank
        self.total_size = int(sum(_ceil_to(int(s), per_round) for s in self.group_sizes))
        self.num_samples = self.total_size // self.num_replicas

    def _plan(self):
        g = self._epoch_generator()
        b, per_round = self.samples_per_gpu, self.samples_per_gpu * self.num_replicas
        parts = [_cycle_to(m[torch.randperm(len(m), generator=g).numpy()], _ceil_to(len(m), per_round))
                 for m in _groups(self.flag)]
        plan = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        plan = _reorder_batches(plan, b, torch.randperm(len(plan) // b, generator=g).numpy())
        lo = self.num_samples * self.rank
        return plan[lo:lo + self.num_samples]
