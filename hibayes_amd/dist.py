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

    def sum_array(self, a):
        """Element-wise sum over ranks of a host array (GEBV partial products of the shards)."""
        t = self.torch.tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def barrier(self):
        self.dist.barrier(group=self.group)


class RcclComm:
    """The library's own RCCL communicator (include/hibayes_gpu.h: hb_comm_*): the per-sweep all-reduce is then ONE
    ncclAllReduce enqueued on the sweep's HIP stream by the library itself — no host synchronisation, no Python in the
    loop. Rank 0 creates the 128-byte RCCL id; `share` ships it (default: a torch.distributed broadcast, which is only used
    for this bootstrap and for the few host-side scalars ibrm() needs)."""

    def __init__(self, rank=0, world=1, device=0, share=None, side=None):
        from ._lib import check, lib
        # _lib.lib() loads torch (where it exists) before the library, so that this library, the system librccl it dlopen()s
        # and torch all use one HIP runtime, and torch's own RCCL build is torn down after ours at exit.
        self.L = lib()
        self.rank, self.world, self.side = int(rank), int(world), side
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(self.L.hb_comm_unique_id(ident))
        if self.world > 1:
            if share is None:
                raise ValueError("RcclComm: world > 1 needs `share` to distribute rank 0's id")
            raw = share(bytes(ident) if self.rank == 0 else None)
            ident = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        check(self.L.hb_comm_init(C.byref(h), ident, self.rank, self.world, int(device)))
        self.handle = h
        if self.L.hb_comm_world(h) != self.world:
            raise RuntimeError("RCCL reports %d ranks, expected %d" % (self.L.hb_comm_world(h), self.world))

    @classmethod
    def from_torch(cls, device_index, group=None):
        """Bootstrap over an initialised torch.distributed process group (any backend)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"
        dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")

        def share(raw):
            t = torch.zeros(128, dtype=torch.uint8, device=dev)
            if raw is not None:
                t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dist.broadcast(t, src=0, group=group)
            return bytes(t.cpu().numpy().tobytes())

        side = TorchComm(device=torch.device("cuda", device_index) if on_gpu else torch.device("cpu"), group=group)
        return cls(rank, world, device_index, share=share, side=side)

    def selftest(self):
        """One small all-reduce with a known answer through the library's communicator (raises if the sum is wrong)."""
        from ._lib import check
        check(self.L.hb_comm_selftest(self.handle))

    def max_int(self, v):
        return int(v) if self.world == 1 else self.side.max_int(v)

    def sum_array(self, a):
        return np.asarray(a, dtype=np.float64) if self.world == 1 else self.side.sum_array(a)

    def barrier(self):
        if self.world > 1:
            self.side.barrier()

    def close(self):
        if getattr(self, "handle", None):
            self.L.hb_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
