"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend nccl == RCCL on ROCm).

The bulk path shards naturally: units are independent and counts are commutative integer sums.
Every rank decodes the same alignment file and keeps the batches whose ordinal is congruent to
its rank (batches hold whole PE pairs, so a pair never straddles two ranks); each GPU holds a
full index replica; the only collective is one sum all-reduce of the u64 count vector (plus the
five log counters) at the end -- ~330 KB for genes_tes, latency-bound on xGMI.
The reference has no distributed path at all.
"""

import os

import numpy as np


class World:
    def __init__(self, rank=0, size=1, local_rank=0, backend=None):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self.backend = backend
        self._own = False

    @classmethod
    def from_env(cls, backend=None):
        size = int(os.environ.get('WORLD_SIZE', 1))
        if size <= 1:
            return cls()
        import torch
        import torch.distributed as dist
        rank = int(os.environ.get('RANK', 0))
        local = int(os.environ.get('LOCAL_RANK', 0))
        w = cls(rank, size, local)
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(local)
                dist.init_process_group(backend, device_id=torch.device('cuda', local))
            else:
                dist.init_process_group(backend)
            w._own = True
        w.backend = dist.get_backend()
        return w

    def mine(self, batch_ordinal):
        """Does this rank count batch number `batch_ordinal` (round-robin sharding)?"""
        return batch_ordinal % self.size == self.rank

    def sum_counts(self, counts, stats_vec):
        """All-reduce (sum) of the per-rank count vector and stats; returns numpy arrays."""
        if self.size == 1:
            return counts, stats_vec
        import torch
        import torch.distributed as dist
        dev = torch.device('cuda', self.local_rank) if self.backend == 'nccl' else torch.device('cpu')
        t = torch.from_numpy(np.concatenate([counts.astype(np.int64), np.asarray(stats_vec, np.int64)])).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t.cpu().numpy()
        return out[:len(counts)].astype(np.uint64), out[len(counts):]

    def close(self):
        if self._own:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
