"""Rank-sharded index streams of the data-parallel input pipeline.

The reference shards training samples with ``DistributedSampler(train_set,
shuffle=True, seed=args.myseed, drop_last=True)`` and re-seeds it every epoch
with ``set_epoch`` (dlib/utils/utils_dataloaders.py:138-148,
dlib/utils/utils_trainer.py:325-326); evaluation with ``shuffle=False,
drop_last=False`` when ``eval_bsize > 1`` (utils_trainer.py:382-386).  This is
the same index law (torch.utils.data.distributed.DistributedSampler, torch
2.x) without the Dataset / DataLoader machinery: the device-side patch
assembly (srhip_patch_gather) consumes plain index lists.  tests/test_cpu_host.py
checks the lists against torch's own sampler for every (n, world, rank, epoch)
tried."""
import math

import torch


class ShardedSampler:
    def __init__(self, n_samples: int, num_replicas: int, rank: int, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        if not 0 <= rank < num_replicas:
            raise ValueError(f"invalid rank {rank} for {num_replicas} replicas")
        self.n, self.num_replicas, self.rank = int(n_samples), int(num_replicas), int(rank)
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, int(seed), drop_last, 0
        if drop_last and self.n % self.num_replicas != 0:
            # the tail that does not divide evenly is dropped
            self.num_samples = math.ceil((self.n - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(self.n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch: int):
        """Every rank draws the SAME permutation from seed + epoch, then takes its stride."""
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad <= len(idx):
                idx += idx[:pad]
            else:
                idx += (idx * math.ceil(pad / len(idx)))[:pad]
        else:
            idx = idx[:self.total_size]
        assert len(idx) == self.total_size
        return idx[self.rank:self.total_size:self.num_replicas]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.num_samples

    def batches(self, batch_size: int, drop_last: bool = True):
        """Index lists of one epoch's minibatches on this rank (DataLoader(batch_size,
        drop_last=True, sampler=...), utils_dataloaders.py:141-148)."""
        idx = self.indices()
        stop = len(idx) - (len(idx) % batch_size if drop_last else 0)
        return [idx[i:i + batch_size] for i in range(0, stop, batch_size)]


def train_sampler(n_samples, world, rank, seed):
    """The reference's training sampler (utils_dataloaders.py:138-139)."""
    return ShardedSampler(n_samples, world, rank, shuffle=True, seed=seed, drop_last=True)


def eval_sampler(n_samples, world, rank):
    """The reference's evaluation sampler (DistributedSampler(shuffle=False), default drop_last=False)."""
    return ShardedSampler(n_samples, world, rank, shuffle=False, drop_last=False)
