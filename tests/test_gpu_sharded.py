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


STAT_KW = dict(niter=1200, nburn=400, thin=5, verbose=False, store_alpha=False)
CASES_SPARSE = (("wide", "BayesCpi", [0.95, 0.05]),)
CASES_BIASED = (("wide", "BayesRR", [0.95, 0.05]), ("ld", "BayesCpi", [0.95, 0.05]), ("ld", "BayesRR", [0.95, 0.05]))


def _worker_stat(rank, world, port, q, cases, seeds=(1, 2, 3), kw=None):
    kw = kw or STAT_KW
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm, shard_range
    comm = TorchComm(device=torch.device("cuda", 0))
    out = {}
    for kind, model, Pi in cases:
        y, X = _stat_data(kind)
        lo, hi = shard_range(X.shape[1], rank, world)
        for seed in seeds:
            f = H.Bayes(y, np.asfortranarray(X[:, lo:hi]), model, Pi, seed=seed, comm=comm, m_global=X.shape[1], m_offset=lo, **kw)
            out[(kind, model, seed)] = (f["Vg"], f["Ve"], f["h2"], f["pi"][0], f["alpha"])
    q.put((rank, out))
    dist.destroy_process_group()


def _sharded_vs_single(world, cases, port0, seeds=(1, 2, 3), kw=None):
    kw = kw or STAT_KW
    ns = len(seeds)
    import torch.multiprocessing as mp
    import hibayes_amd as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port0 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_stat, args=(r, world, port, q, cases, seeds, kw)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=1700) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    rows = []
    for kind, model, Pi in cases:
        y, X = _stat_data(kind)
        one = [H.Bayes(y, X, model, Pi, seed=10 + s, **kw) for s in seeds]
        for k, name in enumerate(("Vg", "Ve", "h2", "pi0")):
            a = np.array([res[0][(kind, model, s)][k] for s in seeds])
            for r in range(1, world):                                                        # replicated on every rank, bit for bit
                assert np.array_equal(a, [res[r][(kind, model, s)][k] for s in seeds])
            b = np.array([(f["Vg"], f["Ve"], f["h2"], f["pi"][0])[k] for f in one])
            se = np.sqrt(a.var(ddof=1) / ns + b.var(ddof=1) / ns)
            rows.append((kind, model, name, a.mean(), b.mean(), se))
        alpha2 = np.mean([np.concatenate([res[r][(kind, model, s)][4] for r in range(world)]) for s in seeds], axis=0)
        alpha1 = np.mean([f["alpha"] for f in one], axis=0)
        rows.append((kind, model, "corr(alpha)", np.corrcoef(alpha1, alpha2)[0, 1], 1.0, 0.0))
    return rows


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world", [2, 8])   # (4 shards ran green in rounds 3 and 4 — profiles/r04_gpu_tests.txt — and cost 160 s of the suite's budget)
def test_sharded_posterior_of_a_sparse_model_is_the_single_gpu_posterior(world):
    """A marker-sharded sweep is not the single-GPU chain (inside a sweep a shard does not see the other shards' moves),
    so it is compared as a sampler of the same posterior: `world` contiguous shards (gloo ranks sharing cuda:0) against one GPU,
    3 seeds each, on data shaped like the configurations the sharded mode is for (BASELINE.json configs 4 and 5: a point-mass
    model, n = 4000 >> markers in the model, independent markers). THE PROPERTY: Vg, Ve, h2, pi agree within max(5 %, 4 Monte-Carlo
    SE), the posterior-mean effects correlate > 0.97, and every rank holds bit-identical replicated quantities — at 2, 4 and 8 shards."""
    # (round 6: eight ranks sharing ONE GPU took 308 s of the suite's 811 — two seeds of 800 sweeps there; the band is 4 SE of what was run)
    seeds, kw = ((1, 2, 3), STAT_KW) if world == 2 else ((1, 2), dict(STAT_KW, niter=800, nburn=300))
    rows = _sharded_vs_single(world, CASES_SPARSE, 31500 + 97 * world, seeds, kw)
    for kind, model, name, a, b, se in rows:
        print("world %d %s %s %s: sharded %.5g vs single %.5g (MC SE %.2g)" % (world, kind, model, name, a, b, se))
        if name == "corr(alpha)":
            assert a > 0.97
        else:
            assert abs(a - b) < max(0.05 * abs(b), 4 * se), (world, name, a, b, se)


@pytest.mark.timeout(1800)
@pytest.mark.xfail(strict=False, reason="measured, documented bias of a partially synchronous sweep where it should NOT be used (DESIGN.md §8), "
                   "2 shards vs 1 GPU: BayesRR on the wide data (all 6000 markers move every sweep) Vg -17.5 % / Ve +30.5 %; the reference's "
                   "demo set (n = 300 < m = 1000, the halves in strong LD) BayesCpi Vg +17.8 % / Ve +22.4 %, BayesRR Vg +10.2 % / Ve +21.2 %")
def test_sharded_posterior_where_every_marker_moves_or_the_shards_are_in_ld():
    rows = _sharded_vs_single(2, CASES_BIASED, 33100)
    bad = []
    for kind, model, name, a, b, se in rows:
        print("%s %s %s: sharded %.5g vs single %.5g (%+.1f %%)" % (kind, model, name, a, b, 100 * (a - b) / abs(b)))
        if name != "corr(alpha)" and abs(a - b) >= max(0.05 * abs(b), 4 * se):
            bad.append((kind, model, name))
    assert not bad, bad


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
def test_sync_every_blocks_lockstep_and_the_measured_effect_on_the_bias():
    """(Not a 'tightening': SURVEY §8e hoped for a monotone knob, the measurement says otherwise — this test pins what IS true.)
    hb_bayes_args.sync_blocks (SURVEY §8e's sync_every_blocks): the shards exchange their residual deltas several times per
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


def _worker_c4(rank, world, port, q, n, m_local):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm
    comm = TorchComm(device=torch.device("cuda", 0))
    m_global, lo = world * m_local, rank * m_local
    with H.Context(n, m_local, m_offset=lo, seed=20240901) as c:
        c.generate(20240901, mono_every=1000)              # columns addressed by GLOBAL marker index: the 2M-marker matrix, this rank's part
        rng = np.random.default_rng(4)                      # the same on every rank
        idx = np.sort(rng.choice(m_global, m_global // 1000, replace=False))
        eff = rng.normal(0, 1, idx.size)
        beta = np.zeros(m_local)
        sel = (idx >= lo) & (idx < lo + m_local)
        beta[idx[sel] - lo] = eff[sel]
        xb = np.zeros(n)
        H._lib.check(c.L.hb_ctx_matvec(c.h, beta.ctypes.data, xb.ctypes.data))
        xb = comm.sum_array(xb)
        xb -= xb.mean()
        xb *= np.sqrt(0.5 / xb.var())
        y = xb + rng.normal(0, np.sqrt(0.5), n)
        c.set_pipeline(1, 2, 4)                            # (a 12-block Gram band: 6 GB per rank instead of 10.7, eight ranks fit with room)
        r = H.Bayes(y, None, "BayesCpi", [0.95, 0.05], niter=4, nburn=0, thin=1, seed=5, verbose=False, comm=comm, m_global=m_global,
                    m_offset=lo, ctx=c, store_alpha=False)
        ra, u = c.get_residual()
        g, trk, _ = c.get_effects()
        xg = np.zeros(n)
        H._lib.check(c.L.hb_ctx_matvec(c.h, g.ctypes.data, xg.ctypes.data))
        xg = comm.sum_array(xg)                             # X g over ALL shards
        mu_last = r["MCMCsamples"]["mu"][0, -1]
        err_u = float(np.abs(u - xg).max() / max(1.0, np.abs(xg).max()))
        err_r = float(np.abs(ra + u - (y - mu_last)).max() / np.abs(y).max())
        q.put((rank, r["mu"], r["Vg"], r["Ve"], r["h2"], float(r["pi"][0]), u[:64].copy(), float(u.sum()), int((g != 0).sum()), err_u, err_r,
               r["timing"]["mean_events"]))
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_eight_rank_dry_run_at_the_shape_of_config_4():
    """BASELINE.json configs[3]: BayesCpi, n = 50k, m = 2M over 8 GPUs — here as 8 ranks of 250 000 markers each on ONE MI355X
    (8 contexts of 12.5 GB genotypes + 6 GB Gram band; gloo carries the per-sweep all-reduce of the residual deltas). Not a
    timing: what an 8-way exchange at that size must keep. Lock-step: every rank ends with bit-identical replicated quantities
    (mu, Vg, Ve, h2, pi, u). The invariant of SURVEY §8e: yadj = y - mu - X g with X g summed over ALL shards, at the sweep
    boundary, to 1e-9. And the shards really worked: thousands of markers entered on every rank in the cold sweeps."""
    import torch
    if torch.cuda.mem_get_info(0)[0] < 175e9:
        pytest.skip("needs ~165 GB of free HBM for eight 250k-marker contexts")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    world = 8
    ps = [ctx.Process(target=_worker_c4, args=(r, world, port, q, 50000, 250000)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted([q.get(timeout=1400) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
    ref = out[0]
    for o in out:
        assert o[1:6] == ref[1:6], "replicated scalars differ on rank %d" % o[0]
        assert np.array_equal(o[6], ref[6]) and o[7] == ref[7]
        assert o[9] < 1e-9 and o[10] < 1e-9, (o[0], o[9], o[10])
        assert o[8] > 1000 and o[11] > 1000
    print("8 ranks x 250k markers: mu %.6f Vg %.5f Ve %.5f h2 %.4f pi0 %.5f; markers in the model per rank %s; max |u - Xg| %.2e, max |yadj + u - (y - mu)| %.2e"
          % (ref[1], ref[2], ref[3], ref[4], ref[5], [o[8] for o in out], max(o[9] for o in out), max(o[10] for o in out)))


def _rows_data():
    rng = np.random.default_rng(31)
    n, m = 900, 1300                                       # shards of 512 + 388 individuals (a 256-multiple and a ragged tail)
    p = rng.uniform(0.05, 0.5, m)
    X = np.asfortranarray((rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8))
    X[:, 9] = 1
    idx = rng.choice(m, 20, replace=False)
    y = X[:, idx].astype(np.float64) @ rng.normal(0, 0.5, 20) + rng.normal(0, 1.0, n)
    return y, X


ROW_MODELS = (("BayesCpi", [0.95, 0.05], None), ("BayesRR", [0.95, 0.05], None), ("BayesR", [0.9375, 0.03125, 0.015625, 0.015625], [0, 1e-4, 1e-3, 1e-2]),
              ("BayesA", [0.95, 0.05], None))
ROW_KW = dict(niter=30, nburn=10, thin=2, seed=77, verbose=False)


def _worker_rows(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hibayes_amd as H
    from hibayes_amd.dist import TorchComm
    comm = TorchComm(device=torch.device("cuda", 0))
    y, X = _rows_data()
    n = y.size
    lo, hi = (0, 512) if rank == 0 else (512, n)
    out = {}
    for model, Pi, fold in ROW_MODELS:
        f = H.Bayes(y[lo:hi], np.asfortranarray(X[lo:hi, :]), model, Pi, fold=fold, comm=comm, shard_rows=True, n_global=n, row_offset=lo, **ROW_KW)
        out[model] = (f["MCMCsamples"]["alpha"], f["Vg"], f["Ve"], f["h2"], f["mu"], f["pi"], f["pip"], f["g"], f["e"])
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_row_sharded_exact_mode_is_the_single_gpu_chain_bit_for_bit():
    """SURVEY §8e's exact alternative, kept as the correctness cross-check mode (hb_bayes_args.shard_rows): the INDIVIDUALS are
    sharded, every panel mat-vec's digit-plane sums — integers — are all-reduced, and every rank runs the whole chain. Two shards
    (512 + 388 individuals, gloo ranks on one MI355X) against one process holding all 900, same mode: BIT FOR BIT — every stored
    effect sample, Vg, Ve, h2, mu, pi, pip; the returned u and e are the shard's rows of the single-process vectors. This holds for the
    models a marker-sharded sweep biases (BayesRR, BayesA: every marker moves) as for the sparse ones. Against the default
    single-GPU path (pipeline, one-workgroup reductions) the same chain to 1e-9: only the order of three n-long sums differs."""
    import torch.multiprocessing as mp
    import hibayes_amd as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    y, X = _rows_data()
    n = y.size
    for model, Pi, fold in ROW_MODELS:
        one = H.Bayes(y, X, model, Pi, fold=fold, shard_rows=True, n_global=n, row_offset=0, **ROW_KW)     # one shard, same code path
        plain = H.Bayes(y, X, model, Pi, fold=fold, **ROW_KW)                                                # the default pipeline
        for rank, (lo, hi) in ((0, (0, 512)), (1, (512, n))):
            a, vg, ve, h2, mu, pi, pip, u, e = res[rank][model]
            assert np.array_equal(a, one["MCMCsamples"]["alpha"]), (model, float(np.abs(a - one["MCMCsamples"]["alpha"]).max()), vg - one["Vg"], mu - one["mu"])
            assert (vg, ve, h2, mu) == (one["Vg"], one["Ve"], one["h2"], one["mu"]), model
            assert np.array_equal(pi, one["pi"]) and np.array_equal(pip, one["pip"])
            assert np.array_equal(u, one["g"][lo:hi])
            np.testing.assert_allclose(e, one["e"][lo:hi], rtol=0, atol=1e-10)       # (X * alpha sums its column blocks with atomics)
        a1, a0 = one["MCMCsamples"]["alpha"], plain["MCMCsamples"]["alpha"]
        assert np.array_equal(a1 != 0, a0 != 0), model
        np.testing.assert_allclose(a1, a0, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose([one["Vg"], one["Ve"], one["h2"], one["mu"]], [plain["Vg"], plain["Ve"], plain["h2"], plain["mu"]], rtol=1e-9)
