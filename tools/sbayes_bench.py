"""SBayesD on the device beside the CPU oracle at a size where the LD matrix matters.
   python tools/sbayes_bench.py [m] [n_ref] [sweeps]  ->  one JSON line per model
Synthetic input: block-LD genotypes (blocks of 64, a column copies its predecessor per individual with probability 0.9), the LD
matrix is their covariance (what ldmat() hands to sbrm()), the summary statistics are the marginal regressions of a simulated trait
on the same individuals (500 causal markers, h2 = 0.5)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hibayes_amd as H  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rng = np.random.default_rng(20240901)

t0 = time.time()
nb = (m + 63) // 64
p = rng.uniform(0.05, 0.5, size=(1, nb * 64)).astype(np.float32)
X = np.empty((n, nb * 64), dtype=np.float32)
X[:, 0::64] = rng.binomial(2, p[:, 0::64], size=(n, nb))
for c in range(1, 64):
    fresh = rng.binomial(2, p[:, (c - 1)::64][:, :nb], size=(n, nb))  # (the block keeps its first column's frequency)
    keep = rng.random((n, nb)) < 0.9
    X[:, c::64] = np.where(keep, X[:, (c - 1)::64], fresh)
X = X[:, :m]
beta = np.zeros(m)
causal = rng.choice(m, size=min(500, m // 20), replace=False)
beta[causal] = rng.normal(size=causal.size)
gv = X @ beta.astype(np.float32)
gv = gv.astype(np.float64)
beta *= np.sqrt(0.5 / gv.var())
y = gv * np.sqrt(0.5 / gv.var()) + rng.normal(scale=np.sqrt(0.5), size=n)
Xc = X - X.mean(axis=0, keepdims=True)
try:
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    Xt = torch.from_numpy(Xc).to(dev).double()
    ldm = (Xt.T @ Xt / (n - 1)).cpu().numpy()
    del Xt
    if dev == "cuda":
        torch.cuda.empty_cache()
except ImportError:
    ldm = (Xc.T.astype(np.float64) @ Xc.astype(np.float64)) / (n - 1)
ldm = np.asfortranarray(ldm)
yc = y - y.mean()
xx = (Xc.astype(np.float64) ** 2).sum(axis=0)
xx[xx == 0] = np.nan
bhat = (Xc.T.astype(np.float64) @ yc) / xx
resid = (yc @ yc - bhat ** 2 * xx) / (n - 2)
sumstat = np.asfortranarray(np.stack([X.mean(axis=0) / 2, bhat, np.sqrt(resid / xx), np.full(m, float(n))], axis=1))
print("# input built in %.1f s: m = %d, LD matrix %.2f GB" % (time.time() - t0, m, ldm.nbytes / 1e9), file=sys.stderr)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as O  # noqa: E402

for model, Pi, fold in (("BayesCpi", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2]), ("BayesRR", [0.95, 0.05], None)):
    r = H.SBayesD(sumstat, ldm, model, Pi, niter=sweeps, nburn=sweeps // 2, thin=5, fold=fold, verbose=False, seed=7, store_alpha=False)
    tg = r["timing"]
    gpu_ms = 1e3 * tg["loop_seconds"] / max(tg["iters_done"], 1)
    ko = 2 if model == "BayesRR" else 4
    t1 = time.time()
    o = O.sbayes(sumstat, ldm, model, Pi, fold=fold, niter=ko, nburn=0, thin=1, seed=7)
    cpu_ms = 1e3 * o["loop_seconds"] / max(o["iters_done"], 1)
    # the same few sweeps on the device: draw for draw the oracle's chain (one Philox stream per marker and iteration)
    rp = H.SBayesD(sumstat, ldm, model, Pi, niter=ko, nburn=0, thin=1, fold=fold, verbose=False, seed=7, store_alpha=False)
    scale = np.abs(o["g_last"]).max() + 1e-300
    gdiff = float(np.abs(rp["g_last"] - o["g_last"]).max() / scale)
    same_set = bool(((rp["g_last"] != 0) == (o["g_last"] != 0)).all())
    moved = tg["mean_events"]
    # bytes a sweep must move: one LD column (m doubles) per marker whose effect changed, read by k_sb_update
    alg = moved * m * 8.0
    print(json.dumps({"path": "SBayesD", "model": model, "m": m, "gpu_ms_per_sweep": round(gpu_ms, 3), "gpu_sweeps_per_s": round(1e3 / gpu_ms, 2),
                      "setup_seconds": round(tg["setup_seconds"], 2), "moves_per_sweep": round(moved, 1),
                      "ld_bytes_per_sweep": alg, "achieved_GBps": round(alg / (gpu_ms * 1e-3) / 1e9, 1),
                      "cpu_oracle_ms_per_sweep": round(cpu_ms, 1), "cpu_sweeps": ko, "gpu_over_cpu": round(cpu_ms / gpu_ms, 1),
                      "h2": round(r["h2"], 4), "parity_sweeps": ko, "same_markers_in_model": same_set, "max_abs_diff_g_over_max_g": gdiff}))
