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


def _stat_data(kind):
    """(y, X) for the sharded-vs-single comparison: 'ld' = the reference's demo set (n = 300 < m = 1000, strong LD between
    the halves); 'wide' = a synthetic set shaped like a real analysis (n >> causal markers, independent markers, n = 4000,
    m = 6000)."""
    import hibayes_amd as H
    if kind == "ld":
        d = os.path.join(ROOT, "tests", "golden", "demo", "demo")
        pl = H.read_plink(d)
        phe = H.read_table(d + ".phe")
        ids = [r[1] for r in pl["fam"]]
        pos = {v: i for i, v in enumerate(phe["id"])}
        rows = [i for i, v in enumerate(ids) if v in pos and phe["T1"][pos[v]] is not None]
        return np.array([float(phe["T1"][pos[ids[i]]]) for i in rows]), np.asfortranarray(pl["geno"][rows, :])
    rng = np.random.default_rng(2024)
    n, m = 4000, 6000
    p = rng.uniform(0.05, 0.5, m)
    X = np.asfortranarray((rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8))
    idx = rng.choice(m, 40, replace=False)
    xb = X[:, idx].astype(np.float64) @ rng.normal(0, 1, 40)
    xb *= np.sqrt(0.5 / xb.var())
    return xb + rng.normal(0, np.sqrt(0.5), n), X


STAT_MODELS = (("BayesCpi", [0.95, 0.05]), ("BayesRR", [0.95, 0.05]))
STAT_KW = dict(niter=2500, nburn=1000, thin=5, verbose=False, store_alpha=False)


def _worker_stat(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm, shard_range
    comm = TorchComm(device=torch.device("cuda", 0))
    out = {}
    for kind in ("wide", "ld"):
        y, X = _stat_data(kind)
        lo, hi = shard_range(X.shape[1], rank, world)
        for model, Pi in STAT_MODELS:
            for seed in (1, 2, 3):
                f = H.Bayes(y, np.asfortranarray(X[:, lo:hi]), model, Pi, seed=seed, comm=comm, m_global=X.shape[1], m_offset=lo, **STAT_KW)
                out[(kind, model, seed)] = (f["Vg"], f["Ve"], f["h2"], f["pi"][0], f["alpha"])
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sharded_posterior_against_the_single_gpu_posterior():
    """A marker-sharded sweep is NOT the single-GPU chain: inside a sweep a shard does not see the other shards' moves
    (SURVEY.md §8 e: "Hogwild"/partially synchronous), so it is compared as a sampler of the same posterior, two shards
    against one GPU, 3 seeds each, BayesCpi (sparse) and BayesRR (every marker moves every sweep).
    * 'wide' data (n = 4000 >> causal markers, independent markers — the shape the sharded mode is meant for), BayesCpi: Vg, Ve,
      h2, pi and the marker effects agree within max(5 %, 4 Monte-Carlo SE). BayesRR on the same data (all 6000 markers in the
      model, m > n): the synchronous update of two dense shards shrinks Vg by ~17 % and inflates Ve by ~30 % — bounded here at
      40 %, documented; hb_bayes_run() warns when a model in which every marker moves (RR / A / L) is sharded.
    * 'ld' data (the reference's demo set: n = 300 < m = 1000, the two halves in strong LD): both shards fit the same signal
      against the same stale residual, the summed update overshoots and the residual variance is biased upwards — measured here
      and bounded, and documented in DESIGN.md §8 as the regime NOT to shard in."""
    import torch.multiprocessing as mp
    import hibayes_amd as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_stat, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=860) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    report = []
    for kind in ("wide", "ld"):
        y, X = _stat_data(kind)
        for model, Pi in STAT_MODELS:
            one = [H.Bayes(y, X, model, Pi, seed=s, **STAT_KW) for s in (11, 12, 13)]
            for k, name in enumerate(("Vg", "Ve", "h2", "pi0")):
                a = np.array([res[0][(kind, model, s)][k] for s in (1, 2, 3)])
                assert np.array_equal(a, [res[1][(kind, model, s)][k] for s in (1, 2, 3)])      # replicated on both ranks
                b = np.array([(f["Vg"], f["Ve"], f["h2"], f["pi"][0])[k] for f in one])
                se = np.sqrt(a.var(ddof=1) / 3 + b.var(ddof=1) / 3)
                rel = (a.mean() - b.mean()) / abs(b.mean()) if b.mean() else 0.0
                report.append("%s %s %s: sharded %.4g vs single %.4g (%+.1f %%, MC SE %.2g)" % (kind, model, name, a.mean(), b.mean(), 100 * rel, se))
                if kind == "wide" and model == "BayesCpi":
                    assert abs(a.mean() - b.mean()) < max(0.05 * abs(b.mean()), 4 * se), report[-1]
                elif kind == "wide":
                    assert abs(rel) < 0.40, report[-1]
                elif name in ("Vg", "Ve", "h2"):
                    assert abs(rel) < 0.5, report[-1]          # biased, but a bounded bias: see the docstring
            alpha2 = np.mean([np.concatenate([res[0][(kind, model, s)][4], res[1][(kind, model, s)][4]]) for s in (1, 2, 3)], axis=0)
            alpha1 = np.mean([f["alpha"] for f in one], axis=0)
            cc = np.corrcoef(alpha1, alpha2)[0, 1]
            report.append("%s %s: correlation of posterior-mean effects %.4f" % (kind, model, cc))
            assert cc > (0.97 if kind == "wide" else 0.7), report[-1]
    print("\n".join(report))


def _worker_sync(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm, shard_range
    comm = TorchComm(device=torch.device("cuda", 0))
    y, X = _stat_data("wide")
    lo, hi = shard_range(X.shape[1], rank, world)
    out = {}
    for blocks in (1, 4):
        for seed in (1, 2):
            f = H.Bayes(y, np.asfortranarray(X[:, lo:hi]), "BayesRR", [0.95, 0.05], seed=seed, comm=comm, m_global=X.shape[1],
                        m_offset=lo, panel=64, sync_every_blocks=blocks, niter=1500, nburn=500, thin=5, verbose=False, store_alpha=False)
            out[(blocks, seed)] = (f["Vg"], f["Ve"], f["g"])
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_every_blocks_tightens_the_sharded_sweep():
    """hb_bayes_args.sync_blocks (SURVEY §8e's sync_every_blocks): the shards exchange their residual deltas several times per
    sweep. BayesRR on the 'wide' data is the case a once-per-sweep exchange biases most (every marker moves, two dense shards
    fit the same residual: Vg about -18 %, Ve about +30 %). Checked: the replicas stay in lock-step, and with 4 exchanges per
    sweep the residual variance — the quantity the stale residual inflates — is within a few per cent of the single-GPU
    posterior (+3 % measured). What the knob does NOT do is remove the bias of a partially synchronous sweep: two blocks
    updated at once against the same residual both fit the signal they share, and Vg goes from -18 % through zero (between 2
    and 3 exchanges on these data) to +23 % at 4 and beyond at 8 (a numpy emulation of the scheme shows the same sign change),
    so the test bounds Vg and DESIGN.md §8 recommends 2 exchanges, 1 for the sparse models."""
    import torch.multiprocessing as mp
    import hibayes_amd as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_sync, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=860) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    y, X = _stat_data("wide")
    one = [H.Bayes(y, X, "BayesRR", [0.95, 0.05], seed=s, panel=64, niter=1500, nburn=500, thin=5, verbose=False, store_alpha=False)
           for s in (11, 12)]
    vg1, ve1 = np.mean([f["Vg"] for f in one]), np.mean([f["Ve"] for f in one])
    bias = {}
    for blocks in (1, 4):
        for seed in (1, 2):
            assert res[0][(blocks, seed)][0] == res[1][(blocks, seed)][0] and res[0][(blocks, seed)][1] == res[1][(blocks, seed)][1]
            assert np.array_equal(res[0][(blocks, seed)][2], res[1][(blocks, seed)][2])          # g = X g_last: replicated
        vg = np.mean([res[0][(blocks, s)][0] for s in (1, 2)])
        ve = np.mean([res[0][(blocks, s)][1] for s in (1, 2)])
        bias[blocks] = ((vg - vg1) / vg1, (ve - ve1) / ve1)
    print("single GPU: Vg %.4g Ve %.4g; sharded, 1 exchange per sweep: %+.1f %% / %+.1f %%; 4 exchanges: %+.1f %% / %+.1f %%" % (
        vg1, ve1, 100 * bias[1][0], 100 * bias[1][1], 100 * bias[4][0], 100 * bias[4][1]))
    assert abs(bias[1][1]) > 0.15                      # the inflated residual variance this knob exists for is really there ...
    assert abs(bias[4][1]) < 0.3 * abs(bias[1][1])     # ... and four exchanges per sweep remove most of it
    assert abs(bias[4][0]) < 0.4                       # Vg: -18 % -> about +23 % (documented sign change), bounded
