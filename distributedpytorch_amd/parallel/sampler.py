"""Rank-sharded sampler — the reference's TODO "# loader sampler distributed"
(`/root/reference/train_pascal.py:3`); its loader used plain ``shuffle=True, drop_last=True``
(`:161`).  Shards are disjoint, cover the (possibly tail-trimmed) dataset, and are equal-sized on
every rank so SyncBatchNorm counts and collective call counts always match."""
from __future__ import annotations

import math
from typing import Iterator, Optional

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


class DistributedSampler(Sampler[int]):
    def __init__(self, dataset_len: int, num_replicas: Optional[int] = None, rank: Optional[int] = None,
                 shuffle: bool = True, seed: int = 0, drop_last: bool = True, pad: bool = True):
        """``pad=False`` (with ``drop_last=False``): no wrap-around padding — every sample is visited exactly once
        over all ranks and shard lengths differ by at most one.  For EVALUATION only (metrics are merged at the
        end, so ranks need not take the same number of steps): duplicated samples would be counted twice in the
        merged confusion matrix / Jaccard sums; the reference evaluates each validation object exactly once
        (`/root/reference/train_pascal.py:237`, `:298`)."""
        if hasattr(dataset_len, "__len__"):
            dataset_len = len(dataset_len)
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if not 0 <= rank < num_replicas:
            raise ValueError("rank out of range")
        self.n, self.replicas, self.rank = int(dataset_len), num_replicas, rank
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.epoch = 0
        self.pad = pad or drop_last
        if drop_last:
            self.per_rank = self.n // num_replicas
        elif self.pad:
            self.per_rank = math.ceil(self.n / num_replicas)
        else:
            self.per_rank = len(range(rank, self.n, num_replicas))
        self.total = self.per_rank * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.per_rank

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed * 1000003 + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if not self.pad:
            return iter(order[self.rank::self.replicas])
        if self.total <= len(order):
            order = order[:self.total]
        else:  # pad by wrapping so every rank gets the same count
            reps = math.ceil(self.total / max(len(order), 1))
            order = (order * reps)[:self.total]
        return iter(order[self.rank:self.total:self.replicas])
