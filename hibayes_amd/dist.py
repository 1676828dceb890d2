"""Marker sharding across the GPUs of one node (SURVEY.md §8 e).

One process per GPU.  Rank p owns the contiguous marker range shard_range(m, p, P) — contiguous
so that LD-correlated neighbours stay on one device — and a full replica of the residual.  Within a
sweep a shard updates only its own replica; once per sweep the shards' residual deltas (and the few
scalar sums the hyper-parameter draws need) are summed with ONE all-reduce over RCCL/xGMI, after
which  yadj = y - mu - ... - X g  holds again on every rank.  The collective itself is
torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests); the library calls back
into `TorchComm` through the hb_allreduce_fn hook of include/hibayes_gpu.h.

The reference has no distributed path at all (single R process); this is new.
"""
import ctypes as C

import numpy as np

from ._lib import ALLREDUCE_FN


def shard_range(m, rank, world):
    """Contiguous, balanced: the first (m % world) ranks get one extra marker."""
    base, extra = divmod(int(m), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class TorchComm:
    """all-reduce(sum) provider on top of an initialised torch.distributed process group."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self._buf = None
        self.calls = 0

    def buffer(self, count):
        if self._buf is None or self._buf.numel() < count:
            self._buf = self.torch.zeros(int(count), dtype=self.torch.float64, device=self.device)
        return self._buf

    def all_reduce_ptr(self, ptr, count):
        buf = self._buf
        if buf is None or ptr != buf.data_ptr() or count > buf.numel():
            return 1
        self.dist.all_reduce(buf[:count], op=self.dist.ReduceOp.SUM, group=self.group)
        if buf.is_cuda:
            self.torch.cuda.synchronize(buf.device)
        self.calls += 1
        return 0

    def make_callback(self, count):
        """Returns (ctypes callback, device pointer of the exchange buffer) for hb_bayes_args."""
        buf = self.buffer(count)

        def _cb(ptr, cnt, _user):
            try:
                return self.all_reduce_ptr(ptr, cnt)
            except Exception:  # never let an exception cross the C boundary
                return 2

        return ALLREDUCE_FN(_cb), buf.data_ptr()

    def max_int(self, v):
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def barrier(self):
        self.dist.barrier(group=self.group)
