"""A sweep whose device pipeline gives up waiting is replayed, and the run is still the oracle's chain.

Every wait of the persistent pipeline is bounded; a waiter that times out raises the abort flag, all kernels of the sweep leave
and hb_ctx_sweep_end() reports HB_ERR_ABORTED. hb_run_step() then puts back the state it saved before the sweep (effects,
residual, u, the posterior counters) and replays it — the per-SNP draws are counter-based (rocRAND Philox addressed by
iteration and marker), so the replay is the same chain; a second failure of the same sweep replays it on the event-ordered
per-panel kernels. The debug hook hb_ctx_debug_inject_abort() raises the flag in mid-sweep, exactly what a waiter that timed out
does. The reference's loop simply runs niter iterations (src/Bayes.cpp:477); a run here must do the same."""
import numpy as np
import pytest

import hibayes_amd as H
from hibayes_amd._lib import HibayesError
from oracle import oracle as O
from test_gpu_depth import geno, pheno, _compare

pytestmark = pytest.mark.gpu

CASES = [  # model, Pi, fold, geometry, panel, columns
    ("BayesRR", [0.95, 0.05], None, (1, 2, 2), 512, 8192),      # k_chain_dense + k_fold_dense + dense update rows
    ("BayesL", [0.95, 0.05], None, (1, 2, 2), 512, 8192),
    ("BayesCpi", [0.95, 0.05], None, (1, 3, 7), 512, 32768),    # k_chain_group + k_fwd
    ("BayesCpi", [0.95, 0.05], None, (1, 2, 8), 512, 32768),    # round 6: eight panels per launch
    ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 1), 512, 16384),  # k_chain_persist
    ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 2), 512, 16384),  # round 6: BayesR on the certified group chain
]


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(20260930)
    X = geno(rng, 2048, 32768)
    return X, pheno(rng, X)


@pytest.mark.parametrize("model,Pi,fold,geo,panel,mcols", CASES)
@pytest.mark.parametrize("at,times", [(3, 1), (9, 1), (5, 2)])
def test_aborted_sweep_is_replayed_as_the_same_chain(data, model, Pi, fold, geo, panel, mcols, at, times):
    X, y = data[0][:, :mcols], data[1]
    m = X.shape[1]
    kw = dict(fold=fold, niter=6, nburn=0, thin=1, seed=424242)
    ref = O.bayes(y, X, model, Pi, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    with H.Context(X.shape[0], m, panel=panel, seed=424242) as c:
        c.upload(X)
        c.set_pipeline(*geo)
        npanels = (m + c.panel - 1) // c.panel
        # the first sweep of the run (it already counts and stores: nburn = 0) is aborted once the chain has published `at` panels;
        # times = 2: its first replay as well, the second replay then runs on the per-panel kernels
        c.debug_inject_abort(min(at, npanels - 1), times)
        r = H.Bayes(y, None, model, Pi, verbose=False, ctx=c, **kw)
        assert c.pipeline()[:3] == geo        # (the geometry is back after a replay on the per-panel kernels)
    assert r["timing"]["sweeps_replayed"] == times
    _compare(r, ref, tol=1e-6 if model == "BayesL" else 1e-9)


def test_without_recovery_an_aborted_sweep_fails_the_run_loudly(data, monkeypatch):
    X, y = data[0][:, :8192], data[1]
    monkeypatch.setenv("HB_RECOVER", "0")
    with H.Context(X.shape[0], X.shape[1], panel=512, seed=1) as c:
        c.upload(X)
        c.set_pipeline(1, 2, 2)
        c.debug_inject_abort(4, 1)
        with pytest.raises(HibayesError) as ei:
            H.Bayes(y, None, "BayesRR", [0.95, 0.05], verbose=False, ctx=c, niter=3, nburn=0, thin=1, seed=1)
        assert ei.value.status == 7 and "timed out" in str(ei.value)
