"""N > 1 plumbing on CPU: torch.distributed (gloo, world_size 2) through the same TorchComm object and
ctypes callback the library calls on the GPU box. The collective is the only data-path exchange of
the sharded sweep (one residual all-reduce per sweep, SURVEY.md §8 e)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hibayes_amd.dist import TorchComm, shard_range
    comm = TorchComm(device=torch.device("cpu"))
    n = 1000
    count = 2 * n + 16
    cb, ptr = comm.make_callback(count)
    buf = comm.buffer(count)
    # what hb_run::step packs: (yadj - yadj0, u - u0, scalar sums); every rank contributes its shard's part
    rng = np.random.default_rng(100 + rank)
    mine = rng.normal(size=count)
    buf.copy_(torch.from_numpy(mine))
    rc = cb(ptr, count, None)           # the library's call: device pointer + count
    bad = cb(ptr + 8, count, None)      # a foreign pointer must be refused, not reduced
    lo, hi = shard_range(10007, rank, world)
    nw = comm.max_int(5 + rank)
    q.put((rank, rc, bad, buf.numpy().copy(), mine, (lo, hi), nw, comm.calls))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_callback_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted([q.get(timeout=100) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=30)
    total = out[0][4] + out[1][4]
    for rank, rc, bad, reduced, mine, rng_, nw, calls in out:
        assert rc == 0 and bad != 0 and calls == 1
        np.testing.assert_allclose(reduced, total, rtol=0, atol=1e-12)   # identical sums on every rank
        assert nw == 6
    assert np.array_equal(out[0][3], out[1][3])  # bitwise identical -> ranks stay in lockstep without a broadcast
    assert out[0][5] == (0, 5004) and out[1][5] == (5004, 10007)
