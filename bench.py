#!/usr/bin/env python
"""bench.py — Gibbs sweeps/sec of the individual-level marker sweep on MI355X.

Metric (BASELINE.json): "Gibbs sweeps/sec (full m-marker pass) + achieved HBM GB/s, n=50k m=500k".
The headline model is BayesCpi (the model BASELINE.json's north_star target is stated for, at n=50k, m=500k);
BayesR (configs[2]) is measured on the same genotypes and reported under "secondary".
One step = one iteration of the reference's MCMC loop (src/Bayes.cpp:477-917): intercept draw,
the full m-marker sweep on the device, the end-of-sweep reductions and the host hyper-parameter
draws — on synthetic genotypes already resident in HBM (SURVEY.md §8 d). `value` is measured on the 2-bit resident layout
(--bits 2) with the library's default mat-vec for that layout (k_dotq2m since round 5); the same invocation also measures
and reports the int8-column layout of SURVEY §8 a1 (`int8`: the HBM-bound v_dot4 kernel north_star's 40 % target is stated for) and the
other 2-bit kernel (`vdot4_ab`: k_dotq2, north_star's literal "no MFMA" formulation).

  python bench.py --gpus 1 --steps K --warmup W            single GPU
  python bench.py --gpus N ...                             starts N ranks itself (torch.distributed.run, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   marker-sharded, launcher supplied

Weak scaling: every rank holds --m markers (m_global = N * m); `value` counts passes over m markers
by all ranks per second, i.e. N * (global sweeps/s).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# arithmetic of the dominant kernel (everything else — chain, residual, updates — is f64 in every mode)
DTYPE = {0: "f32", 1: "f64", 2: "i8 x i8 -> i32 (exact fixed point: the f64 residual as 7 int8 digit planes; error below an f64 ddot's)"}

LINE_LIMIT = 6000   # bytes; the driver keeps ~10 KB of stdout tail and parses the last line out of it (round 5's 23 KB line was lost)


def _r(x, sig=6):
    """numbers of the compact line: 6 significant digits"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except (TypeError, ValueError):
        return None


def _leg(b, name):
    """one side leg on the compact line: {leg, model, bits, kernel, value, ms_per_step, frac (mat-vec launches in situ), frac_sweep (resident bytes x sweeps/s
    over the HBM peak; the all-move models: both passes), frac_of_measured_copy, regime}"""
    if not isinstance(b, dict):
        return None
    if "error" in b:
        return {"leg": name, "model": b.get("model"), "error": str(b["error"])[:80]}
    rf = b.get("roofline", {})
    return {"leg": name, "model": b.get("model"), "bits": b.get("resident_genotype_bits"), "kernel": rf.get("kernel"),
            "value": _r(b.get("value")), "ms_per_step": _r(b.get("ms_per_step")), "frac": _r(rf.get("frac"), 4), "frac_sweep": _r(b.get("achieved_frac_of_hbm_peak"), 4),
            "frac_of_measured_copy": _r(rf.get("frac_of_measured_copy"), 4), "regime": b.get("regime")}


def compact_line(res, full_path):
    """The ONE JSON line the driver parses: the contract's top-level keys, a config of five keys, ONE roofline block and ONE
    cpu_baseline block of scalars, and one small object per side leg. Everything else (in-situ statistics, regime curves, states,
    notes) is in the full record written to `full_path`. Kept under LINE_LIMIT bytes (asserted here and in tests/test_host_logic.py)."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(res.get(k)) for k in top}
    cfg = res.get("config", {})
    line["config"] = {"workload": str(cfg.get("workload", ""))[:160], "model": cfg.get("model"), "n": cfg.get("n"), "m": cfg.get("m_per_gpu"),
                      "m_global": cfg.get("m_global"), "bits": cfg.get("resident_genotype_bits")}
    rf = res.get("roofline", {})
    line["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms",
                                                   "launches_per_sweep", "measured_copy_GBps", "frac_of_measured_copy")}
    line["roofline"]["isolated_frac"] = _r(rf.get("isolated", {}).get("frac"), 4)
    line["regime"] = res.get("regime")
    line["achieved_GBps"] = _r(res.get("achieved_GBps"))                        # the metric's second half: resident genotype bytes x sweeps/s, all ranks
    line["achieved_frac_of_hbm_peak"] = _r(res.get("achieved_frac_of_hbm_peak"), 4)
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        if "error" in cb:
            line["cpu_baseline"] = {"error": str(cb["error"])[:120]}
        else:
            line["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "value_1thread", "int8_value", "int8_cores", "host_cpu_quota")}
            line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:100]
            if cb.get("value"):
                line["vs_cpu_baseline"] = _r(res["value"] / cb["value"], 4)
    legs = []
    for key, name in (("int8", "int8"), ("vdot4_ab", "vdot4"), ("mfma_ab", "mfma"), ("secondary", "secondary")):
        lg = _leg(res.get(key), name)
        if lg:
            legs.append(lg)
        conv = res.get(key, {}).get("converged") if isinstance(res.get(key), dict) else None
        if isinstance(conv, dict):
            legs.append(_leg(conv, name + "_converged"))
    for b in res.get("all_move", []) or []:
        legs.append(_leg(b, "all_move"))
    line["legs"] = legs
    if res.get("n_gpus", 1) > 1:
        pr = res.get("per_rank_ms_per_step") or {}
        line["per_rank_ms_per_step"] = {"min": _r(pr.get("min")), "max": _r(pr.get("max"))}
        ar = res.get("allreduce") or {}
        line["allreduce"] = {"ms_per_call": _r(ar.get("ms_per_call_back_to_back")), "bytes": ar.get("bytes"), "collective": str(cfg.get("collective", ""))[:90]}
        line["ranks_counted_by_all_reduce"] = res.get("ranks_counted_by_all_reduce")
        st = res.get("strong")
        if isinstance(st, dict):
            line["strong_value"] = _r(st.get("value"))
            line["strong"] = {k: _r(st.get(k)) for k in ("m_global", "m_per_gpu", "ms_per_step", "model", "error") if k in st}
    line["full"] = full_path
    out = json.dumps(line, separators=(",", ":"))
    if len(out) >= LINE_LIMIT:      # never lose the record to its own size again: drop the legs, keep the contract
        line["legs"] = [{"leg": lg.get("leg"), "model": lg.get("model"), "value": lg.get("value"), "frac": lg.get("frac")} for lg in legs if lg]
        out = json.dumps(line, separators=(",", ":"))
    assert len(out) < LINE_LIMIT, "bench line is %d bytes" % len(out)
    return out


def write_full(res):
    """the full record (everything round 5 printed on the line) beside the compact line: profiles/bench_last_full.json, and
    gpurun_out/ when that exists (it is what travels back from a gpurun box)"""
    path = os.path.join("profiles", "bench_last_full.json")
    for d in ("profiles", "gpurun_out"):
        try:
            if d == "gpurun_out" and not os.path.isdir(os.path.join(ROOT, d)):
                continue
            with open(os.path.join(ROOT, d, "bench_last_full.json"), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            pass
    return path


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--burnin", type=int, default=400,
                    help="MCMC burn-in sweeps run as part of the set-up, before the warm-up steps: the chain starts with ~5 %% of the "
                         "markers in the model and needs a few hundred sweeps to reach the regime a 20 000-iteration run spends its time in (round 5: 400, "
                         "not 300 — the markers changed per sweep still fall from 950 to 770 between sweep 300 and 450, and with --warmup 5 the timed "
                         "sweeps sat on that slope)")
    ap.add_argument("--burnin-secondary", type=int, default=300)
    ap.add_argument("--burnin-converged", type=int, default=0,
                    help="the secondary model (BayesR) is measured twice: after --burnin-secondary sweeps (the chain has not found the signal "
                         "yet: ~30 000 markers in the model, ~49 000 moves per sweep) and again after this many MORE sweeps, continued from "
                         "the first leg's state (nnz ~2 000-5 000: the regime a 50 000-iteration run lives in); 0 (default since round 6: the leg is "
                         "43 of the bench's 57 seconds) = first figure only; 2500 = round 5's converged leg")
    ap.add_argument("--secondary-state", default=os.path.join("profiles", "state", "bayesr_config3.npz"),
                    help="a stored chain state of the secondary model on THIS synthetic data (sparse effects + hyper-parameters after 2 800 sweeps; written "
                         "by --save-state, checked against n / m / seed / model): the `converged` leg continues from it (hb_warm_state, 40 untimed sweeps first) "
                         "instead of spending 40 s on its own burn-in; used when --burnin-converged is 0 and the file matches")
    ap.add_argument("--init-state", default="", help="start the HEADLINE model's chain from a stored state (.npz written by --save-state for that model on this data) instead of "
                                                     "the prior defaults: profiling runs of a converged chain (tools/r6_profiles.sh)")
    ap.add_argument("--save-state", default="", help="write the secondary model's state after its converged leg (needs --burnin-converged > 0) to this .npz")
    ap.add_argument("--stamped", type=int, default=10,
                    help="sweeps run right after the timed region with every block of every mat-vec launch stamped on the device's "
                         "100 MHz clock (hb_ctx_matvec_stamps): the in-situ launch duration the roofline is computed from")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (gloo: development runs on one GPU)")
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--m", type=int, default=int(os.environ.get("HB_BENCH_M", "500000")),
                    help="markers per GPU (under torch.distributed.run pass it as HB_BENCH_M: the launcher claims --m)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --m markers per GPU (m_global = N * m); strong: --m-global markers over all GPUs "
                         "(BASELINE.json configs[3]: 2 000 000 over 8)")
    ap.add_argument("--m-global", type=int, default=int(os.environ.get("HB_BENCH_M_GLOBAL", "2000000")))
    ap.add_argument("--collective", default="rccl", choices=["rccl", "torch"],
                    help="per-sweep residual all-reduce: rccl = ncclAllReduce enqueued by the library itself on the sweep stream "
                         "(hb_comm_*); torch = torch.distributed callback (host-synchronous; also the fall-back)")
    ap.add_argument("--model", default="BayesCpi", help="BASELINE.json north_star target: BayesCpi at n=50k, m=500k")
    ap.add_argument("--secondary", default="BayesR", help="second model measured on the same genotypes ('' = none)")
    ap.add_argument("--tertiary", default="BayesRR,BayesA,BayesL",
                    help="the models in which every marker moves every sweep (comma-separated), each reported in the 'all_move' list ('' = none)")
    ap.add_argument("--no-strong", action="store_true", help="--gpus > 1, weak scaling: skip the strong-scaling leg (config 4: --m-global markers over all ranks)")
    ap.add_argument("--no-ab", action="store_true", help="skip the two side legs of the headline model (matrix-core A/B kernel, int8 columns)")
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--precise", type=int, default=2,
                    help="panel mat-vec arithmetic: 2 = exact fixed point (7 int8 digit planes of the fp64 residual, int32 dot4 "
                         "accumulation; fp64-grade, the library default), 1 = fp64 FMA, 0 = fp32 image of the residual")
    ap.add_argument("--bits", type=int, default=int(os.environ.get("HB_BENCH_BITS", "2")), choices=[2, 8],
                    help="resident genotype layout the sweep reads: 8 = int8 columns (SURVEY §8 a1), 2 = 2 bits per genotype (§8 f1: "
                         "PLINK's density, expanded in registers inside the mat-vec; a quarter of the bytes; same chain bit for bit)")
    ap.add_argument("--matvec-kernel", type=int, default=int(os.environ.get("HB_DOTQ2_KIND", "2")), choices=[0, 1, 2],
                    help="2-bit layout: which kernel computes the panel mat-vec for the headline `value` — 2 = k_dotq2m (matrix cores; the library "
                         "default since round 5), 0 = k_dotq2 (v_dot4), 1 = k_dotq2r; the other of {2, 0} is measured as the `vdot4_ab` / `mfma_ab` side leg")
    ap.add_argument("--seed", type=int, default=20240901)
    ap.add_argument("--cpu-m", type=int, default=8000, help="markers of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-sweeps", type=int, default=4)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check (no GPU needed): the ranks rendezvous, count each other with one all-reduce and rank 0 prints "
                         "a JSON line with n_gpus = the world size; nothing is measured")
    return ap.parse_args()


def synth_phenotype(ctx, n, m, m_offset, m_global, seed, comm, model):
    """y = X beta + e with h2 = 0.5 and 0.1% causal markers (SURVEY.md §8 d); BayesR: the three
    non-null classes in ratio 2:2:1 with variances 1e-4 : 1e-3 : 1e-2."""
    rng = np.random.default_rng(seed + 17)  # identical on every rank
    n_causal = max(1, int(round(0.001 * m_global)))
    idx = np.sort(rng.choice(m_global, n_causal, replace=False))
    eff = rng.normal(0.0, 1.0, n_causal)
    if model == "BayesR":
        cls = rng.choice(3, n_causal, p=[0.4, 0.4, 0.2])
        eff *= np.sqrt(np.array([1e-4, 1e-3, 1e-2])[cls])
    beta = np.zeros(m)
    loc = (idx >= m_offset) & (idx < m_offset + m)
    beta[idx[loc] - m_offset] = eff[loc]
    xb = np.zeros(n)
    from hibayes_amd._lib import check
    check(ctx.L.hb_ctx_matvec(ctx.h, beta.ctypes.data, xb.ctypes.data))
    if comm is not None:
        t = comm.torch.tensor(xb, device=comm.device)
        comm.dist.all_reduce(t)
        xb = t.cpu().numpy()
    xb -= xb.mean()
    xb *= np.sqrt(0.5 / xb.var())
    return xb + rng.normal(0.0, np.sqrt(0.5), n)


def cpu_baseline(ctx, y, args, Pi, fold, g_warm=None):
    """Times oracle/hb_oracle.c (the faithful port of src/Bayes.cpp: double column-major X, serial
    marker loop, BLAS-1 shaped dot/axpy) on the first --cpu-m markers of the same data, then scales
    to m markers (cost per sweep is exactly linear in m: one independent column per marker).
    g_warm: the GPU chain's effects after its timed region — the port starts from them, i.e. in the same
    stationary regime (markers in the model = daxpy count) the GPU figure is quoted in."""
    from oracle import oracle as O
    mc = min(args.cpu_m, args.m)
    X8 = np.asfortranarray(ctx.download(0, mc))
    Xd = np.asfortranarray(X8, dtype=np.float64)  # the reference's layout: 8 bytes per genotype
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    out = {}
    it = 1 + args.cpu_sweeps
    # nburn = niter - 1: sweeps run exactly as in the reference loop, one stored record
    gi = None if g_warm is None else np.ascontiguousarray(g_warm[:mc])
    r = O.bayes(y, Xd, args.model, Pi, fold=fold, niter=it, nburn=it - 1, thin=1, threads=1,
                rng=O.RNG_PHILOX, seed=args.seed, g_init=gi)
    out[1] = r["iters_done"] / r["loop_seconds"] * (mc / float(args.m))
    one_thread_s = r["loop_seconds"]
    # threaded dot/axpy (what a threaded BLAS would give the reference, README.md:18): ONE persistent team of
    # workers for the whole run, each owning a fixed chunk of the rows, partial sums through padded slots and a two-level tree
    # (oracle/hb_oracle.c team_*; rounds 1-4 forked an OpenMP region per call and ran slower with more threads). Each setting
    # is bounded by a wall-clock guard in a child process.
    import multiprocessing as mp

    def timed(thr, Xm=None):
        Xm = Xd if Xm is None else Xm

        def _child(q):
            rr = O.bayes(y, Xm, args.model, Pi, fold=fold, niter=it, nburn=it - 1, thin=1, threads=thr,
                         rng=O.RNG_PHILOX, seed=args.seed, g_init=gi)
            q.put(rr["iters_done"] / rr["loop_seconds"])

        q = mp.get_context("fork").Queue()
        p = mp.get_context("fork").Process(target=_child, args=(q,))
        p.start()
        limit = max(20.0, 3.0 * one_thread_s)
        p.join(timeout=limit)
        if p.is_alive():
            p.kill()      # (SIGKILL: a team of spinning OpenMP threads does not always honour SIGTERM in time)
            p.join()
            return "slower than %.0f s for %d sweeps (%.1f s on one thread): stopped" % (limit, it, one_thread_s)
        return q.get() * (mc / float(args.m)) if not q.empty() else "no result"

    slow = {}
    # the container's CPU-time quota (cgroup cpu.max): a team with more runnable (spinning) threads than the quota allows is throttled by the
    # scheduler — on the round-5 box (2 x EPYC 9575F, 256 logical CPUs visible, quota 16) 32 threads still ran, 64 ran ten times slower
    # than one — so the thread counts stop at the quota
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    cap = cores if quota is None else max(1, min(cores, int(quota)))
    for thr in sorted(set([min(cap, 4), min(cap, 8), min(cap, 16), min(cap, 64)]) - {1}):
        v = timed(thr)
        if isinstance(v, float):
            out[thr] = v
        else:
            slow[str(thr)] = v
    best_thr = max(out, key=lambda k: out[k])
    # SURVEY 8d, last sentence: the same sampler on int8 columns (NOT the reference's layout: reported separately, never as "the baseline")
    i8 = {}
    for thr in sorted({1, best_thr}):
        v = timed(thr, X8)
        if isinstance(v, float):
            i8[thr] = v
    i8_best = max(i8, key=lambda k: i8[k]) if i8 else None
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": out[best_thr], "unit": "sweeps/s", "cores": best_thr, "kind": "port",
            "sample": "oracle/hb_oracle.c (double col-major X, serial marker loop, dot/axpy on a persistent team of row-chunk workers) on the first %d of %d "
                      "markers, n=%d, %d sweeps, scaled by m_sample/m" % (mc, args.m, args.n, 1 + args.cpu_sweeps),
            "value_1thread": out[1], "int8_value": i8.get(i8_best), "int8_cores": i8_best, "int8_by_threads": {str(k): v for k, v in sorted(i8.items())},
            "int8_what": "the same port reading int8 columns (1 byte per genotype instead of the reference's 8): not the reference's layout, reported separately",
            "by_threads": {str(k): v for k, v in sorted(out.items())}, "not_finished": slow,
            "regime": "warm start from the GPU chain's effects after its timed region" if gi is not None else "cold start",
            "host_cores": cores, "host_cpu_quota": quota,
            "cpu_model": cpu_model}


# measured on this chip (tools/dot4_rate.hip, profiles/r04_dot4_rate.txt): a wave64 VALU instruction of the k_dotq2 loop (v_dot4_i32_i8,
# v_and, v_lshrrev) issues once per this many cycles per SIMD, at this sustained clock; LDS returns 128 bytes per clock per CU
VALU_CYCLES_PER_WAVE_INSTR = 4.5
SUSTAINED_GHZ = 2.25
N_CU, SIMD_PER_CU, LDS_BYTES_PER_CLK_CU = 256, 4, 256.0   # (ds_read_b128: 4 cycles per wave-instruction, MI355X_MICROARCH.md LDS table)


def valu_block(n, cols, avg_ms, kernel):
    """What bounds the 2-bit v_dot4 kernel (it is not HBM): per 16 genotypes of a column one lane issues 28 v_dot4 + 5 mask
    operations (round 4: shift-free masks, hb_dotq2.hpp Q2_SCALED) and receives 7 broadcast digit reads + a quarter of a 16-byte
    column read through LDS."""
    if kernel != "k_dotq2":
        return None
    lane_steps = float(n) * cols / 16.0          # (lane, 16-genotype chunk) pairs per launch
    wave_steps = lane_steps / 64.0
    dot4, mask, lds_reads = 28.0 * wave_steps, 5.0 * wave_steps, 7.25 * wave_steps
    t_valu = (dot4 + mask) * VALU_CYCLES_PER_WAVE_INSTR / (N_CU * SIMD_PER_CU) / (SUSTAINED_GHZ * 1e9)
    t_lds = lds_reads * 1024.0 / (N_CU * LDS_BYTES_PER_CLK_CU) / (SUSTAINED_GHZ * 1e9)
    dur = avg_ms * 1e-3
    return {"wave_instructions_per_launch": {"v_dot4_i32_i8": dot4, "mask": mask, "ds_read_b128": lds_reads},
            "cycles_per_wave_instruction_per_simd": VALU_CYCLES_PER_WAVE_INSTR, "sustained_clock_ghz": SUSTAINED_GHZ,
            "valu_issue_floor_us": t_valu * 1e6, "lds_return_floor_us": t_lds * 1e6, "measured_us": dur * 1e6,
            "frac_of_valu_issue_bound": t_valu / dur, "frac_of_lds_return_bound": t_lds / dur,
            "what": "floors of one launch if nothing else existed: VALU issue of the dot4 + mask stream over 1024 SIMDs, and the LDS->VGPR "
                    "return path (1 KiB per ds_read_b128 at 256 B/clk/CU) of the broadcast digit reads; one wave issues both, so where they "
                    "do not overlap the bound is their sum "
                    "(profiles/r04_dot4_rate.txt, profiles/r04_pmc_sq_k_dotq2.txt)"}


def roofline_block(args, n, cols, launches, insitu, iso_ms, traffic, bits=8, kind=0, copy_gbps=None):
    """roofline of the dominant kernel (the panel mat-vec), PHYSICAL: achieved = the bytes of genotypes the resident layout holds for
    one launch (n x columns x bits / 8) / the average duration of the sweep's full-width mat-vec launches AS THE SWEEP RUNS THEM
    (device-clock stamps of every block, chain workgroup and update rows beside them); frac = achieved / the 8 TB/s HBM peak.
    `genotypes_priced_at_one_byte` keeps SURVEY 8d's n x m denominator (one byte per genotype whatever the layout) for comparison
    with the int8 line; `isolated` is the same launch shape replayed without update rows and chain between two HIP events."""
    one = float(n) * cols                # SURVEY §8 d's algorithmic bytes at one byte per genotype
    res = one * bits / 8.0               # bytes the kernel really has to pull from HBM
    avg_ms = insitu["avg_ms"] if insitu else iso_ms
    ach = res / (avg_ms * 1e-3) / 1e9
    kernel = ("k_dotq2m" if kind == 2 else "k_dotq2r" if kind == 1 else "k_dotq2") if (bits == 2 and args.precise == 2) else ("k_dotq" if args.precise == 2 else "k_dot")
    bound = "hbm" if kernel in ("k_dotq", "k_dot", "k_dotq2m") else "valu"
    # (k_dotq2m: priced against HBM — what its launch would be bound by if nothing else were; in situ its update rows wait for the chain
    # workgroup, isolated it is limited by how fast 64-column tiles of 512 individuals arrive per compute unit: see DESIGN.md section 6)
    r = {"bound": bound, "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "bytes_per_launch": res, "avg_launch_ms": avg_ms,
         "launches_per_sweep": insitu["launches_per_sweep"] if insitu else launches, "columns_per_launch": cols,
         "resident_bits_per_genotype": bits,
         "measured": "in situ: device-clock stamps of every block of every mat-vec launch over %d sweeps following the timed region"
                     % insitu["sweeps"] if insitu else "isolated replay (HIP events)",
         "isolated": {"avg_launch_ms": iso_ms, "achieved": res / (iso_ms * 1e-3) / 1e9, "frac": res / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "what": "the same launches without update rows and chain, graph replay between two HIP events"},
         "genotypes_priced_at_one_byte": {"bytes_per_launch": one, "achieved": one / (avg_ms * 1e-3) / 1e9, "frac": one / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                          "what": "SURVEY 8d's n x m denominator: genotypes per second priced at one byte each, whatever "
                                                  "the resident layout; equals `frac` for int8 columns, not a bandwidth for 2-bit ones"}}
    if copy_gbps:
        # "also report fraction of a measured streaming-read kernel on the same buffer" — what the chip delivers when the
        # resident genotypes are only read (hb_ctx_time_stream_read), so that "achievable" is visible beside the nominal 8 TB/s
        r["measured_copy_GBps"] = copy_gbps
        r["frac_of_measured_copy"] = ach / copy_gbps
    if bound == "valu":
        r["bound_note"] = ("the 2-bit v_dot4 kernel is bound by LDS return bandwidth and VALU issue (see `valu`), not by HBM: `frac` is the HBM "
                           "roofline of the bytes it moves and is far below 1 by construction")
        r["valu"] = valu_block(n, cols, avg_ms, kernel)
    if insitu:
        r["in_situ"] = {k: insitu[k] for k in ("min_ms", "max_ms", "sum_ms", "span_ms", "blocks_per_launch", "full_width_launches",
                                               "ms_per_step_of_the_stamped_sweeps")}
    return r


def copy_ceiling(ctx, note=None):
    """GB/s of a plain streaming read of the resident genotype buffer (one untimed pass, then 3 timed)"""
    try:
        ms, nb = ctx.time_stream_read(reps=3)
        gbps = nb / (ms * 1e-3) / 1e9
        if note:
            note("streaming read of the resident genotypes: %.1f MB in %.3f ms = %.0f GB/s" % (nb / 1e6, ms, gbps))
        return gbps
    except Exception:  # noqa: BLE001  (a reported side number)
        return None


def regime_tag(ms_per_step, insitu):
    """A leg whose timed ms per step differs by more than 10 % from the stamped sweeps that follow it was not timed in the regime
    the roofline's launches ran in (markers still being re-admitted or shed): tagged, not hidden."""
    if not insitu:
        return {"regime": "unchecked (no stamped sweeps)"}
    st = insitu["ms_per_step_of_the_stamped_sweeps"]
    rel = abs(ms_per_step - st) / max(1e-12, st)
    return {"regime": "stationary" if rel <= 0.10 else "transient", "timed_vs_stamped_ms_per_step": [ms_per_step, st]}


def pmc_traffic(n, cols, kernel):
    """HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE passes kept under profiles/ (counters cannot be read from
    inside this process): reported only when a pass with the same kernel, n and launch width exists."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")))
        for e in table["passes"]:
            if e["kernel"] == kernel and e["n"] == n and e["columns_per_launch"] == cols:
                return e["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


# untimed sweeps at the start of a side leg of the headline model (matrix-core A/B, int8 columns), whatever --warmup says: the leg
# continues from the headline run's effects AND hyper-parameters (hb_warm_state), these sweeps only let the geometry choice, the hot
# list and the row cache settle (round 4: the legs restarted from prior-default pi / varg and re-admitted tens of thousands of markers
# — 76 instead of 209 sweeps/s under the driver's --warmup 5)
SIDE_WARMUP = 40

PIPELINE = {  # (pipeline, look-ahead groups, panels per mat-vec launch): see DESIGN.md §2
    "BayesCpi": (1, 2, 7), "BayesC": (1, 2, 7), "BayesB": (1, 2, 7), "BayesBpi": (1, 2, 7),
    "BayesR": (1, 2, 2), "BayesRR": (1, 2, 2), "BayesA": (1, 2, 2), "BayesL": (1, 2, 2),   # (RR / A / L at panel 512: k_chain_dense, hb_run's own choice; BayesR: (2, 2) = the certified group chain, with the geometry chosen by regime — (2, 1) / k_chain_persist while many markers move)
}


for _k in list(PIPELINE):  # (tuning runs: HB_BENCH_GEO_BayesR="1,3,1" overrides a model's geometry)
    if os.environ.get("HB_BENCH_GEO_" + _k):
        PIPELINE[_k] = tuple(int(x) for x in os.environ["HB_BENCH_GEO_" + _k].split(","))


def prior(model):
    if model == "BayesR":
        return [0.95, 0.02, 0.02, 0.01], [0.0, 1e-4, 1e-3, 1e-2]  # R/bayes.r:273-275
    return [0.95, 0.05], None


def measure(H, L, ctx, y, model, K, W, args, rank, local_rank, world, m_offset, m_global, comm, torch, note, burn=0, g_init=None, warm=None):
    """burn set-up sweeps, W warm-up iterations, then exactly K iterations between barriers; returns the result dict pieces.
    g_init + warm: effects and hyper-parameters (hb_warm_state) the chain starts from — a leg that CONTINUES in the regime an
    earlier leg reached (same markers in the model, same pi / varg / vare) instead of re-admitting markers from the prior
    defaults. measure.final = (effects, WarmState) after the leg's last sweep."""
    measure.insitu = None
    from hibayes_amd._lib import BayesArgs, RunInfo, WarmState, check
    n, m = args.n, args.m
    Pi, fold = prior(model)
    a = BayesArgs()
    a.n, a.m = n, m
    yv = np.ascontiguousarray(y)
    a.y = yv.ctypes.data
    a.model = model.encode()
    pv = np.array(Pi)
    a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold is not None:
        fv = np.array(fold)
        a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = burn + W + K + args.stamped + 8, 0, 5  # every sweep counts PIP, every 5th is a stored record
    a.outfreq, a.verbose = 0, 0
    a.seed, a.device, a.precise, a.store_alpha = args.seed, local_rank, args.precise, 0
    a.ctx = ctx.h
    keep = []
    if g_init is not None:
        gi = np.ascontiguousarray(g_init, dtype=np.float64)
        a.g_init = gi.ctypes.data
        keep.append(gi)
    if warm is not None:
        a.warm = ct.addressof(warm)
        keep.append(warm)
    if comm is not None:
        a.rank, a.world, a.m_global, a.m_offset = rank, world, m_global, m_offset
        if getattr(comm, "rccl", None) is not None:
            a.comm = comm.rccl.handle
        else:
            cb, ptr = comm.make_callback(L.hb_exchange_count(n))
            a.allreduce, a.exchange_buf = cb, ptr
            keep.append(cb)
    run = ct.c_void_p()
    check(L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32()

    def sync():
        torch.cuda.synchronize(local_rank)
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize(local_rank)

    curve = []
    if burn > 0:
        # the burn-in doubles as the regime curve: sweeps/s against markers changed per sweep, from the cold start (5 % of
        # the markers enter in the first sweeps) down to the stationary regime the timed region runs in
        done, prev = 0, RunInfo()
        check(L.hb_run_state(run, ct.byref(prev)))
        for upto in sorted(set(min(burn, x) for x in (5, 20, 60, 150, burn))):
            if upto <= done:
                continue
            sync()
            tc = time.perf_counter()
            check(L.hb_run_step(run, upto - done, ct.byref(fin)))
            sync()
            dt = time.perf_counter() - tc
            cur = RunInfo()
            check(L.hb_run_state(run, ct.byref(cur)))
            mv = (cur.mean_events * cur.iter - prev.mean_events * prev.iter) / max(1, upto - done)
            curve.append({"sweeps": "%d-%d" % (done, upto), "moves_per_sweep": round(mv, 1), "sweeps_per_s": round((upto - done) / dt, 2)})
            done, prev = upto, cur
        note("%s: burn-in done (%d sweeps)" % (model, burn))
    check(L.hb_run_step(run, W, ct.byref(fin)))
    sync()
    note("%s: warm-up done" % model)
    info0 = RunInfo()
    check(L.hb_run_state(run, ct.byref(info0)))
    t1 = time.perf_counter()
    check(L.hb_run_step(run, K, ct.byref(fin)))
    sync()
    elapsed = time.perf_counter() - t1
    note("%s: timed region done: %.3fs for %d steps" % (model, elapsed, K))
    measure.per_rank_ms = None
    if comm is not None:
        mine = torch.zeros(world, dtype=torch.float64, device=comm.device)
        mine[rank] = elapsed
        comm.dist.all_reduce(mine)                      # one slot per rank: every rank's own time for its K steps
        measure.per_rank_ms = [float(v) / K * 1e3 for v in mine.cpu()]
        elapsed = float(mine.max().item())              # the contract: MAX over ranks
    info = RunInfo()
    check(L.hb_run_state(run, ct.byref(info)))
    # ---- in-situ duration of the dominant kernel: the SAME run goes on for a few sweeps with every block of every mat-vec
    # launch stamped (device clock, 100 MHz); chain workgroup and update rows beside them exactly as in the timed region ----
    insitu = None
    if args.stamped > 0 and args.precise == 2:
        ctx.set_profiling(8)
        check(L.hb_run_step(run, 1, ct.byref(fin)))     # (re-captures the sweep with the stamp pointers: untimed)
        sync()
        acc = {"avg_ms": 0.0, "sum_ms": 0.0, "span_ms": 0.0, "min_ms": 1e9, "max_ms": 0.0}
        t2 = time.perf_counter()
        wall = 0.0
        for _ in range(args.stamped):
            tw = time.perf_counter()
            check(L.hb_run_step(run, 1, ct.byref(fin)))
            sync()
            wall += time.perf_counter() - tw
            st = ctx.matvec_stamps()
            for k in ("avg_ms", "sum_ms", "span_ms"):
                acc[k] += st[k] / args.stamped
            acc["min_ms"], acc["max_ms"] = min(acc["min_ms"], st["min_ms"]), max(acc["max_ms"], st["max_ms"])
        insitu = dict(acc, launches_per_sweep=st["launches_all"], full_width_launches=st["launches"], blocks_per_launch=st["blocks"],
                      columns_per_launch=st["cols_per_launch"], sweeps=args.stamped, ms_per_step_of_the_stamped_sweeps=wall / args.stamped * 1e3)
        ctx.set_profiling(0)
        note("%s: %d stamped sweeps: k_dotq %.2f us per launch in situ (%d launches, stream span %.3f ms, sweep %.3f ms)"
             % (model, args.stamped, acc["avg_ms"] * 1e3, st["launches_all"], acc["span_ms"], wall / args.stamped * 1e3))
    measure.insitu = insitu
    measure.replayed = info.sweeps_replayed
    info_end = RunInfo()
    check(L.hb_run_state(run, ct.byref(info_end)))
    g_end, _, vl_end = ctx.get_effects()
    measure.final = (g_end, WarmState.from_info(info_end, len(Pi), vl_end if model == "BayesL" else None))
    measure.state = {"NumNZSnp": info_end.nnz, "pi": [info_end.pi[j] for j in range(len(Pi))], "varg": info_end.varg,
                     "vare": info_end.vare, "vara": info_end.vara}
    L.hb_run_destroy(run)
    del keep
    ev = (info.mean_events * info.iter - info0.mean_events * info0.iter) / max(1, K)   # over the timed sweeps only
    ms = (info.mean_misses * info.iter - info0.mean_misses * info0.iter) / max(1, K)
    measure.redo = (info.mean_redo * info.iter - info0.mean_redo * info0.iter) / max(1, K)   # speculative chain rounds rolled back and repeated, per sweep
    measure.curve = curve
    return elapsed, ev, info.nnz, ms


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU, exactly as the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` does;
    rank 0's JSON line is the child's stdout, i.e. ours. (--m travels as HB_BENCH_M: see parse().)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
        elif a == "--m":
            skip = True
        elif not a.startswith("--m="):
            argv.append(a)
    env = dict(os.environ, HB_BENCH_M=str(args.m), MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    print("[bench] --gpus %d without a launcher: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = None
    import torch
    if args.dry_run:
        seen = 1
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(backend="gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_counted_by_all_reduce": seen, "gpus_asked_for": args.gpus,
                              "config": {"collective": "gloo all_reduce over %d ranks (launcher check only)" % world}}))
        return
    if world > 1:
        import torch.distributed as dist
        local_rank = local_rank % max(1, torch.cuda.device_count())  # (several ranks on one GPU: --backend gloo only)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
        from hibayes_amd.dist import TorchComm, RcclComm
        comm = TorchComm(device=torch.device("cuda", local_rank))
        comm.rccl, comm.rccl_note = None, "torch.distributed all_reduce (%s) over %d ranks through the library's callback" % (args.backend, world)
        if args.collective == "rccl" and args.backend == "nccl":
            # the library's own communicator; torch.distributed only ships rank 0's RCCL id. ncclCommInitRank is a blocking
            # collective: it runs in a helper thread with a deadline, so that a bootstrap that never completes on this node
            # costs 90 s and the tested torch.distributed path, not the whole run
            import threading
            box = {}

            def _init():
                try:
                    rc = RcclComm.from_torch(local_rank)
                    rc.selftest()          # one all-reduce with a known answer, still under the deadline
                    box["comm"] = rc
                except Exception as e:  # noqa: BLE001
                    box["err"] = e

            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("HB_RCCL_INIT_TIMEOUT", "90")))
            if th.is_alive():
                comm.rccl_note += " (in-library RCCL: ncclCommInitRank did not return in time)"
            elif "comm" in box:
                comm.rccl = box["comm"]
                comm.rccl_note = "ncclAllReduce inside libhibayes_gpu on the sweep stream, %d ranks" % comm.rccl.L.hb_comm_world(comm.rccl.handle)
            else:
                comm.rccl_note += " (in-library RCCL unavailable: %r)" % (box.get("err"),)
            ok = comm.max_int(0 if comm.rccl is not None else 1)   # all ranks or none
            if ok != 0 and comm.rccl is not None:
                comm.rccl.close()
                comm.rccl = None
    import hibayes_amd as H
    L = H.lib()

    n, m = args.n, args.m
    m_global, m_offset = m * world, m * rank
    if args.scaling == "strong" and world > 1:
        from hibayes_amd.dist import shard_range
        m_global = args.m_global
        m_offset, hi = shard_range(m_global, rank, world)
        m = args.m = hi - m_offset

    def note(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    t_start = t0 = time.time()
    ctx = H.Context(n, m, device=local_rank, panel=args.panel, precise=args.precise, m_offset=m_offset,
                    seed=args.seed)
    ctx.generate(args.seed, mono_every=1000)
    gen_s = time.time() - t0
    note("genotypes generated on device (%.2fs)" % gen_s)
    y = synth_phenotype(ctx, n, m, m_offset, m_global, args.seed, comm, args.model)
    note("phenotype built")
    geo = PIPELINE.get(args.model, (1, 1, 1))
    if geo == (1, 3, 7) and args.bits != 2:
        geo = (1, 2, 7)  # int8 columns: the sweep is as long as the HBM-bound mat-vec stream, the third group of look-ahead buys nothing (205 vs 200 sweeps/s)
    ctx.set_pipeline(*geo)
    adaptive = (geo in ((1, 2, 7), (1, 3, 7), (1, 2, 8)) or (args.model == "BayesR" and geo == (1, 2, 2))) and not os.environ.get("HB_NO_ADAPTIVE")
    if adaptive:
        ctx.set_adaptive(True)  # narrow band while many markers move (burn-in), this geometry once few do (the timed region)
    gram_s = ctx.build_gram()
    note("Gram blocks built (%.2fs)" % gram_s)
    pack_s = 0.0
    if args.bits == 2 and args.precise == 2:
        t0 = time.time()
        ctx.set_layout(2, keep_int8=False)   # packed on the device from the int8 columns, which are then dropped
        pack_s = time.time() - t0
        note("genotypes packed to 2 bits, int8 copy dropped (%.2fs)" % pack_s)
    bits = ctx.layout()[0]
    kind_main = args.matvec_kernel if (bits == 2 and args.precise == 2) else 0
    if bits == 2 and args.precise == 2:
        ctx.set_matvec_kernel(kind_main)

    K, W = args.steps, args.warmup
    g_start = warm_start = None
    if args.init_state:
        from hibayes_amd._lib import WarmState
        z = np.load(args.init_state)
        assert (int(z["n"]), int(z["m"]), int(z["seed"]), str(z["model"])) == (n, m, args.seed, args.model), "--init-state: not this model / data"
        g_start = np.zeros(m)
        g_start[z["idx"]] = z["val"]
        warm_start = WarmState.make(float(z["mu"]), float(z["vare"]), float(z["varg"]), [float(x) for x in z["pi"]])
    elapsed, mean_events, nnz, misses = measure(H, L, ctx, y, args.model, K, W, args, rank, local_rank, world, m_offset,
                                        m_global, comm, torch, note, burn=args.burnin, g_init=g_start, warm=warm_start)
    replayed_main = getattr(measure, "replayed", 0)
    redo_main = getattr(measure, "redo", None)
    per_rank_main = getattr(measure, "per_rank_ms", None)
    g_main, warm_main = measure.final     # the state the side legs of this model continue from
    state_main = dict(measure.state)
    curve_main = list(getattr(measure, "curve", []))
    curve_main.append({"sweeps": "timed region", "moves_per_sweep": round(mean_events, 1), "sweeps_per_s": round(world * K / elapsed, 2)})
    insitu_main = measure.insitu
    # the dominant kernel on its own (kept beside the in-situ figure as roofline.isolated): the sweep's mat-vec launches without
    # update rows and without the chain, back to back, HIP events on their stream
    ctx.time_matvec(reps=1)                                   # (untimed: clocks and TLBs as in the steady state of a run)
    iso_ms, launches, cols = ctx.time_matvec(reps=5)
    kernel_main = (("k_dotq2m" if kind_main == 2 else "k_dotq2r" if kind_main == 1 else "k_dotq2") if bits == 2 else "k_dotq") if args.precise == 2 else "k_dot"
    copy_main = copy_ceiling(ctx, note)
    roof = roofline_block(args, n, cols, launches, insitu_main, iso_ms, pmc_traffic(n, cols, kernel_main), bits, kind_main, copy_main)
    note("mat-vec timing pass done")

    # one unit = one pass over m_ref markers (the metric's m = 500k); all ranks together pass over m_global markers per step
    m_ref = float(os.environ.get("HB_BENCH_M", "500000")) if args.scaling == "weak" else 500000.0
    value = (K / elapsed) * (m_global / (float(m) if args.scaling == "weak" else m_ref))
    Pi, fold = prior(args.model)
    res = {
        "metric": "Gibbs sweeps/sec (full m-marker pass) + achieved HBM GB/s, n=50k m=500k",
        "value": value, "unit": "sweeps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": DTYPE[args.precise], "data": "synthetic",
        "config": {"workload": "%s marker sweep, n=%d individuals x m=%d markers per GPU (m_global=%d), genotypes resident at %d bits, "
                               "panel=%d, pipeline=%s" % (args.model, n, m, m_global, bits, ctx.panel, (geo,)),
                   "model": args.model, "n": n, "m_per_gpu": m, "m_global": m_global, "panel": ctx.panel,
                   "pipeline": {"persistent_chain": geo[0], "lookahead_groups": geo[1], "panels_per_matvec": geo[2],
                                "geometry_by_regime": bool(adaptive),
                                "geometry_of_the_timed_region": list(ctx.pipeline())},
                   "sharding": "markers, contiguous ranges, one residual all-reduce per sweep" if world > 1 else "none",
                   "collective": comm.rccl_note if comm is not None else "none",
                   "mcmc_burn_in_sweeps_before_warmup": args.burnin,
                   "mean_changed_markers_per_sweep": mean_events, "row_cache_misses_per_sweep": misses, "NumNZSnp_last": nnz,
                   "chain_rounds_rolled_back_per_sweep": redo_main,
                   "sweeps_replayed_after_a_device_time_out": replayed_main,
                   "matvec_kernel": roof["kernel"],
                   "resident_genotype_bits": bits, "setup_seconds": {"generate": gen_s, "gram": gram_s, "pack_2bit": pack_s}},
        # PHYSICAL bytes: the genotypes the resident layout holds are read once per sweep by the mat-vec (the Gram rows, digit planes and
        # residual updates of a stationary BayesCpi sweep add < 2 %); the n x m-priced figure of earlier rounds is `genotype_rate`
        "achieved_GBps": (K / elapsed) * n * m_global * bits / 8 / 1e9,
        "achieved_frac_of_hbm_peak": (K / elapsed) * n * m_global * bits / 8 / 1e9 / (HBM_PEAK_GBPS * world),
        "genotype_rate": {"value": (K / elapsed) * n * m_global / 1e9, "unit": "G genotypes/s",
                          "what": "n x m genotypes per sweep (SURVEY 8d's unit; = GB/s at one byte per genotype)"},
        "roofline": roof,
        "regime_curve": curve_main,   # sweeps/s is set by the serial chain, i.e. by how many markers change per sweep
        "state_after": state_main,
    }
    res.update(regime_tag(elapsed / K * 1e3, insitu_main))
    roof["regime"] = res["regime"]
    roof["traffic_source"] = ("separate rocprofv3 --pmc FETCH_SIZE pass of the same launch shape, read from the tracked profiles/r05_pmc_traffic.json "
                              "(counters cannot be collected inside this process)") if roof.get("traffic") is not None else "none (no counter pass on file for this launch shape)"

    copy_by_bits = {bits: copy_main}

    def leg_block(model, el, Kx, Wx, ev, nnzx, missx, ins, iso, launches_x, cols_x, bits_x, kind, geo_x, burn_x, curve=None):
        if bits_x not in copy_by_bits:
            copy_by_bits[bits_x] = copy_ceiling(ctx, note)     # (the leg's layout is the resident one while leg_block runs)
        kern = roofline_block(args, n, cols_x, launches_x, ins, iso, None, bits_x, kind, copy_by_bits[bits_x])
        kern["traffic"] = pmc_traffic(n, cols_x, kern["kernel"])
        b = {"model": model, "value": Kx / el, "unit": "sweeps/s", "steps": Kx, "warmup": Wx, "ms_per_step": el / Kx * 1e3,
             "resident_genotype_bits": bits_x, "roofline": kern,
             "achieved_GBps": Kx / el * n * m * bits_x / 8 / 1e9, "achieved_frac_of_hbm_peak": Kx / el * n * m * bits_x / 8 / 1e9 / HBM_PEAK_GBPS,
             "mcmc_burn_in_sweeps_before_warmup": burn_x,
             "mean_changed_markers_per_sweep": ev, "row_cache_misses_per_sweep": missx, "NumNZSnp_last": nnzx,
             "sweeps_replayed_after_a_device_time_out": getattr(measure, "replayed", 0),
             "pipeline": {"persistent_chain": geo_x[0], "lookahead_groups": geo_x[1], "panels_per_matvec": geo_x[2]}}
        b.update(regime_tag(el / Kx * 1e3, ins))
        b["chain_rounds_rolled_back_per_sweep"] = getattr(measure, "redo", None)
        b["state_after"] = dict(getattr(measure, "state", {}))
        if curve is not None:
            b["regime_curve"] = curve
        return b

    single = world == 1
    # ---- A/B: the same sweep with the OTHER 2-bit mat-vec kernel (headline on the matrix-core k_dotq2m: the v_dot4 k_dotq2, north_star's literal
    # "no MFMA" formulation and the default until round 4; headline on k_dotq2: k_dotq2m). Same exact integers, same chain. ----
    ab_key, ab_kind = ("vdot4_ab", 0) if kind_main == 2 else ("mfma_ab", 2)
    if single and bits == 2 and args.precise == 2 and not args.no_ab:
        try:
            ctx.set_matvec_kernel(ab_kind)
            Wm = SIDE_WARMUP
            elm, evm, nnzm, missm = measure(H, L, ctx, y, args.model, K, Wm, args, rank, local_rank, world, m_offset, m_global, comm, torch, note,
                                            burn=0, g_init=g_main, warm=warm_main)
            insm = measure.insitu
            ctx.time_matvec(reps=1)
            isom, lm, cm = ctx.time_matvec(reps=3)
            res[ab_key] = leg_block(args.model, elm, K, Wm, evm, nnzm, missm, insm, isom, lm, cm, 2, ab_kind, geo, 0)
            res[ab_key]["note"] = ("A/B, not the headline: the same sweep with the 2-bit mat-vec on %s; same exact integers and the same chain; continued from the "
                                   "headline run's effects and hyper-parameters" % ("k_dotq2 (v_dot4_i32_i8, lane = column: no MFMA)" if ab_kind == 0 else
                                                                                     "k_dotq2m (v_mfma_i32_16x16x64_i8: the seven digit planes as a skinny int8 GEMM)"))
        except Exception as e:
            res[ab_key] = {"error": repr(e)}
        ctx.set_matvec_kernel(kind_main)
    # ---- the int8-column layout of SURVEY 8 a1 (north_star's own layout, the library's default): HBM-bound k_dotq ----
    if single and bits == 2 and args.precise == 2 and not args.no_ab:
        try:
            geo8 = (1, 2, 7) if geo == (1, 3, 7) else geo
            ctx.set_layout(8)
            ctx.set_pipeline(*geo8)
            W8 = SIDE_WARMUP
            el8, ev8, nnz8, miss8 = measure(H, L, ctx, y, args.model, K, W8, args, rank, local_rank, world, m_offset, m_global, comm, torch, note,
                                            burn=0, g_init=g_main, warm=warm_main)
            ins8 = measure.insitu
            ctx.time_matvec(reps=1)
            iso8, l8, c8 = ctx.time_matvec(reps=3)
            res["int8"] = leg_block(args.model, el8, K, W8, ev8, nnz8, miss8, ins8, iso8, l8, c8, 8, 0, geo8, 0)
            res["int8"]["note"] = ("the same model on int8 columns (SURVEY 8 a1; hb_bayes_args.genotype_bits = 0, the library default), "
                                   "warm-started from the headline run's effects: the mat-vec is HBM-bound here and `roofline.frac` is the fraction of "
                                   "the 8 TB/s peak north_star's 40 % target refers to")
        except Exception as e:
            res["int8"] = {"error": repr(e)}
    sides = [(args.secondary, "secondary", args.burnin_secondary, max(10, K // 4), max(5, min(W, 30)))]
    for tm in [x for x in args.tertiary.split(",") if x]:
        sides.append((tm, "all_move", 20, max(10, K // 8), 5))
    for side, key, burn_s, K2, W2 in sides:
        if not side or side == args.model or world != 1 or (key == "all_move" and side == args.secondary):
            continue
        try:  # a side model is a reported side number: its failure must not take the headline line with it
            # the other model families of BASELINE.json's configs on the same genotypes, shorter runs
            geo2 = PIPELINE.get(side, (1, 1, 1))
            ctx.set_pipeline(*geo2)
            bits2 = ctx.layout()[0]
            if bits2 == 2 and side in ("BayesRR", "BayesA", "BayesL"):
                # the models in which every marker moves: their dense update rows read int8 columns (hb_kernels.hip: dense_upd). BayesR stays on the
                # 2-bit layout since the end of round 6 (k_dotq2m beside its chains: 64.5 / 102.8 against 57.8 / 98.3 sweeps/s cold / converged) — which is
                # also what a run picks by itself (genotype_bits = 0)
                ctx.set_layout(8)
                bits2 = 8
            ctx.build_gram()
            if args.bits == 2 and args.precise == 2 and side not in ("BayesRR", "BayesA", "BayesL") and ctx.layout()[0] != 2:
                ctx.set_layout(2, keep_int8=True)   # (the int8 leg before this one left the int8 columns resident; the all-move legs need them again)
                ctx.set_matvec_kernel(kind_main)
            bits2 = ctx.layout()[0]
            y2 = synth_phenotype(ctx, n, m, m_offset, m_global, args.seed, comm, side)
            el2, ev2, nnz2, miss2 = measure(H, L, ctx, y2, side, K2, W2, args, rank, local_rank, world, m_offset,
                                     m_global, comm, torch, note, burn=burn_s)
            ins2 = measure.insitu
            ctx.time_matvec(reps=1)
            iso2, launches2, cols2 = ctx.time_matvec(reps=2)
            curve2 = list(getattr(measure, "curve", []))
            curve2.append({"sweeps": "timed region", "moves_per_sweep": round(ev2, 1), "sweeps_per_s": round(K2 / el2, 2)})
            blk = leg_block(side, el2, K2, W2, ev2, nnz2, miss2, ins2, iso2, launches2, cols2, bits2, kind_main if bits2 == 2 else 0, tuple(ctx.pipeline()[:3]), burn_s, curve2)
            if key == "all_move":
                blk["note"] = ("every marker moves every sweep: a sweep reads the genotypes twice (mat-vec and residual update), 2 n m bytes — "
                               "`achieved_GBps` prices both passes, `roofline` the mat-vec launches (whose update rows ride in them)")
                blk["achieved_GBps"] *= 2.0
                blk["achieved_frac_of_hbm_peak"] *= 2.0
                res.setdefault(key, []).append(blk)
            else:
                res[key] = blk
                st_path = os.path.join(ROOT, args.secondary_state) if args.secondary_state else ""
                st = None
                if args.burnin_converged == 0 and st_path and os.path.exists(st_path):
                    z = np.load(st_path)
                    if (int(z["n"]), int(z["m"]), int(z["seed"]), str(z["model"])) == (n, m, args.seed, side) and len(z["pi"]) == len(prior(side)[0]):
                        st = z
                if st is not None:
                    from hibayes_amd._lib import WarmState
                    g3 = np.zeros(m)
                    g3[st["idx"]] = st["val"]
                    warm3 = WarmState.make(float(st["mu"]), float(st["vare"]), float(st["varg"]), [float(x) for x in st["pi"]])
                    el3, ev3, nnz3, miss3 = measure(H, L, ctx, y2, side, K2, SIDE_WARMUP, args, rank, local_rank, world, m_offset, m_global, comm, torch, note,
                                                    burn=0, g_init=g3, warm=warm3)
                    ins3 = measure.insitu
                    ctx.time_matvec(reps=1)       # (the geometry by regime may have changed the launch width since the first leg: (2, 1) cold, (2, 2) converged)
                    iso3, launches3, cols3 = ctx.time_matvec(reps=2)
                    blk["converged"] = leg_block(side, el3, K2, SIDE_WARMUP, ev3, nnz3, miss3, ins3, iso3, launches3, cols3, bits2, kind_main if bits2 == 2 else 0, tuple(ctx.pipeline()[:3]), int(st["sweeps"]))
                    blk["converged"]["note"] = ("continued (effects + hyper-parameters, hb_warm_state) from the state stored in %s: this model on this synthetic data after %d sweeps "
                                                "— the chain has found the signal, few markers are left in the model; %d untimed sweeps first"
                                                % (args.secondary_state, int(st["sweeps"]), SIDE_WARMUP))
                if args.burnin_converged > 0:
                    g2, warm2 = measure.final
                    el3, ev3, nnz3, miss3 = measure(H, L, ctx, y2, side, K2, W2, args, rank, local_rank, world, m_offset, m_global, comm, torch, note,
                                                    burn=args.burnin_converged, g_init=g2, warm=warm2)
                    curve3 = list(getattr(measure, "curve", []))
                    curve3.append({"sweeps": "timed region", "moves_per_sweep": round(ev3, 1), "sweeps_per_s": round(K2 / el3, 2)})
                    blk["converged"] = leg_block(side, el3, K2, W2, ev3, nnz3, miss3, measure.insitu, iso2, launches2, cols2, bits2, kind_main if bits2 == 2 else 0, geo2,
                                                 burn_s + W2 + K2 + args.stamped + 1 + args.burnin_converged, curve3)
                    blk["converged"]["note"] = ("the same run continued (effects + hyper-parameters, hb_warm_state) for %d more sweeps: the chain has "
                                                "found the signal, few markers are left in the model" % args.burnin_converged)
                    if args.save_state and rank == 0:
                        g4, w4 = measure.final
                        nz = np.flatnonzero(g4)
                        os.makedirs(os.path.dirname(os.path.abspath(args.save_state)), exist_ok=True)
                        np.savez_compressed(args.save_state, n=n, m=m, seed=args.seed, model=side, idx=nz.astype(np.int32), val=g4[nz], mu=w4.mu, vare=w4.vare,
                                            varg=w4.varg, pi=np.array([w4.pi[j] for j in range(len(prior(side)[0]))]),
                                            sweeps=burn_s + 2 * (W2 + K2 + args.stamped + 1) + args.burnin_converged)
                        note("state of %s written to %s (%d markers in the model)" % (side, args.save_state, nz.size))
                    blk["note"] = ("measured %d sweeps after a cold start, where the chain has not found the signal yet (tens of thousands of markers "
                                   "in the model, see state_after); `converged` is the later regime" % burn_s)
        except Exception as e:
            if key == "all_move":
                res.setdefault(key, []).append({"model": side, "error": repr(e)})
            else:
                res[key] = {"model": side, "error": repr(e)}
    if world > 1:
        # what the scaling curve is made of: every rank's own time per step, and the exchange a sweep ends with on its own
        res["per_rank_ms_per_step"] = {"min": min(per_rank_main), "max": max(per_rank_main), "all": per_rank_main} if per_rank_main else None
        try:
            xc = int(L.hb_exchange_count(n))
            buf = torch.zeros(xc, dtype=torch.float64, device=comm.device)
            for _ in range(5):
                comm.dist.all_reduce(buf)
            torch.cuda.synchronize(local_rank)
            ta = time.perf_counter()
            for _ in range(50):
                comm.dist.all_reduce(buf)
            torch.cuda.synchronize(local_rank)
            ar_ms = (time.perf_counter() - ta) / 50 * 1e3
            res["allreduce"] = {"doubles": xc, "bytes": xc * 8, "ms_per_call_back_to_back": ar_ms, "calls_per_sweep": 1,
                                "share_of_ms_per_step": ar_ms / (elapsed / K * 1e3),
                                "what": "torch.distributed all_reduce (%s) of the sweep's exchange message on its own, 50 calls back to back; "
                                        "inside a sweep the library enqueues the same collective on the sweep stream" % args.backend}
        except Exception as e:  # noqa: BLE001
            res["allreduce"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            res["cpu_baseline"] = cpu_baseline(ctx, y, args, Pi, fold, g_main)
        except Exception as e:  # the baseline is a reported side number, never the measured path
            res["cpu_baseline"] = {"error": repr(e)}
    ctx.close()
    if world > 1:
        try:
            ones = torch.ones(1, dtype=torch.float64, device=comm.device)
            comm.dist.all_reduce(ones)
            res["ranks_counted_by_all_reduce"] = int(ones.item())
        except Exception as e:  # noqa: BLE001
            res["ranks_counted_by_all_reduce"] = repr(e)
    if world > 1 and args.scaling == "weak" and not args.no_strong:
        # BASELINE.json configs[3] in the SAME invocation (the driver passes no extra flags): --m-global markers (2 000 000) sharded
        # over the ranks, the headline model, 2-bit resident — `strong_value` in passes over 500 000 markers per second
        try:
            from hibayes_amd.dist import shard_range
            mg = args.m_global
            lo, hi = shard_range(mg, rank, world)
            ms_ = hi - lo
            cs = H.Context(n, ms_, device=local_rank, panel=args.panel, precise=args.precise, m_offset=lo, seed=args.seed)
            cs.generate(args.seed, mono_every=1000)
            m_keep, args.m = args.m, ms_
            try:
                ys = synth_phenotype(cs, n, ms_, lo, mg, args.seed, comm, args.model)
                cs.set_pipeline(*geo)
                if adaptive:
                    cs.set_adaptive(True)
                cs.build_gram()
                if args.bits == 2 and args.precise == 2:
                    cs.set_layout(2, keep_int8=False)
                    cs.set_matvec_kernel(kind_main)
                els, evs, nnzs, _ = measure(H, L, cs, ys, args.model, K, W, args, rank, local_rank, world, lo, mg, comm, torch, note, burn=args.burnin)
            finally:
                args.m = m_keep
                cs.close()
            res["strong"] = {"value": (K / els) * (mg / 500000.0), "unit": "passes over 500 000 markers per second, all ranks together",
                             "model": args.model, "m_global": mg, "m_per_gpu": ms_, "ms_per_step": els / K * 1e3, "steps": K, "warmup": W,
                             "global_sweeps_per_s": K / els, "mean_changed_markers_per_sweep_per_rank": evs, "NumNZSnp_last": nnzs,
                             "per_rank_ms_per_step": getattr(measure, "per_rank_ms", None),
                             "what": "BASELINE.json configs[3]: BayesCpi, n=50k, m=2M sharded over the ranks, one residual all-reduce per sweep"}
        except Exception as e:  # noqa: BLE001  (a side leg: its failure must not take the line with it)
            res["strong"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(compact_line(res, write_full(res)), flush=True)
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
