"""Rank partition of the training set: the semantics of the reference's ``DistIterSampler``
(mmsr/data/data_sampler.py:8-69) and of its batch rule (mmsr/data/__init__.py:70-73).

Every rank draws the SAME epoch-seeded permutation of ``total_size = ceil(len(dataset) * ratio / world) * world`` slots
(``torch.randperm`` on a CPU generator seeded with the epoch), folds it onto the dataset with ``% len(dataset)`` and keeps
the slots ``rank, rank + world, rank + 2 world, ...``: the ranks' index lists are disjoint slices of one permutation, equally
long, and their union is the whole enlarged epoch.  Nothing is communicated -- which is why the data path of the multi-GPU
step has no collective; the only exchange of a training step is DDP's gradient all-reduce (base_model.py:70-72).
"""
import math

import torch
from torch.utils.data.sampler import Sampler


class DistIterSampler(Sampler):
    """``DistIterSampler(dataset, num_replicas=None, rank=None, ratio=100)``: same constructor, ``__iter__``, ``__len__``
    and ``set_epoch`` as the reference class (:31-69); ``num_replicas`` / ``rank`` default to the process group's."""

    def __init__(self, dataset, num_replicas=None, rank=None, ratio=100):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError('DistIterSampler: pass num_replicas and rank, or initialise torch.distributed first')
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError(f'rank {rank} outside [0, {num_replicas})')
        self.dataset = dataset
        self.num_replicas = int(num_replicas)
        self.rank = int(rank)
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) * ratio / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def epoch_slots(self):
        """The whole enlarged epoch (every rank's share interleaved): ``randperm(total_size, seed = epoch) % len(dataset)``."""
        g = torch.Generator()
        g.manual_seed(self.epoch)
        return torch.randperm(self.total_size, generator=g) % len(self.dataset)

    def __iter__(self):
        mine = self.epoch_slots()[self.rank::self.num_replicas]
        assert mine.numel() == self.num_samples
        return iter(mine.tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


def per_rank_batch_size(global_batch, world_size):
    """``batch_size`` of the YAML is the batch over ALL GPUs; each rank loads ``batch_size // world`` and the reference
    asserts divisibility (data/__init__.py:70-73) -- stage 3's 9 does not divide by 8, BASELINE configs[3] uses 32."""
    if global_batch % world_size != 0:
        raise AssertionError(f'batch_size {global_batch} is not a multiple of the world size {world_size} '
                             '(mmsr/data/__init__.py:72)')
    return global_batch // world_size
