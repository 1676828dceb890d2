"""The marker-sharded sampler end to end on the GPU: two ranks (gloo; both on cuda:0, RCCL needs one device per rank) run
hb_bayes_run on their halves of the markers with one residual all-reduce per sweep (SURVEY.md §8 e). A sharded chain is not
the single-GPU chain, so the checks are the invariants it must keep: identical replicated state on both ranks, and
yadj = y - mu - X alpha-path consistency (g returned = last-iteration u = X_global g_last, the sum of the shards' parts)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm, shard_range
    comm = TorchComm(device=torch.device("cuda", 0))
    rng = np.random.default_rng(7)                       # the same global data on every rank
    n, m = 600, 1500
    p = rng.uniform(0.05, 0.5, m)
    X = ((rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8))
    beta = np.zeros(m); idx = rng.choice(m, 15, replace=False); beta[idx] = rng.normal(0, 0.4, 15)
    y = X @ beta + rng.normal(0, 1.0, n)
    lo, hi = shard_range(m, rank, world)
    r = H.Bayes(y, np.asfortranarray(X[:, lo:hi]), "BayesCpi", [0.95, 0.05], niter=60, nburn=20, thin=5, seed=99, verbose=False,
                comm=comm, m_global=m, m_offset=lo)
    q.put((rank, (lo, hi), r["alpha"], r["g"], r["e"], r["mu"], r["Vg"], r["Ve"], r["h2"], r["pi"], X, y))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_run_keeps_the_replicas_in_lockstep():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted([q.get(timeout=280) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=30)
    (r0, s0, a0, g0, e0, mu0, vg0, ve0, h0, pi0, X, y), (r1, s1, a1, g1, e1, mu1, vg1, ve1, h1, pi1, _, _) = out
    assert s0 == (0, 750) and s1 == (750, 1500)
    # replicated quantities are bitwise identical: every rank draws the hyper-parameters from the same host stream and sees
    # the same all-reduced residual
    assert np.array_equal(g0, g1) and mu0 == mu1 and vg0 == vg1 and ve0 == ve1 and h0 == h1 and np.array_equal(pi0, pi1)
    assert np.array_equal(e0, e1)
    # e = y - mu - X alpha with alpha the posterior mean over ALL markers (each rank holds its shard of alpha)
    alpha = np.concatenate([a0, a1])
    np.testing.assert_allclose(e0, y - mu0 - X.astype(float) @ alpha, rtol=0, atol=1e-8)
    assert np.count_nonzero(alpha) > 0 and 0.0 < h0 < 1.0
