"""Offline model of k_pre's BayesR boundary search (hb_pre.hpp bayesr_threshold): evaluations of h per boundary before and after round 6's
stopping rule, and how far the roots move. python tools/r6_newton_sim.py"""
import numpy as np
rng = np.random.default_rng(0)
K = 4
fold = np.array([0, 1e-4, 1e-3, 1e-2])


def params(xx, pi, varg, vare):
    a, b = np.zeros(K), np.zeros(K)
    a[0] = np.log(pi[0])
    for c in range(1, K):
        vf = varg * fold[c]
        vv = xx + vare / vf
        a[c] = -0.5 * np.log(vf * xx / vare + 1) + np.log(pi[c])
        b[c] = 0.5 / (vv * vare)
    return a, b


def h(q, a, b, c, logT):
    s = a + b * q
    mA, mB = s[:c + 1].max(), s[c + 1:].max()
    wA, wB = np.exp(s[:c + 1] - mA), np.exp(s[c + 1:] - mB)
    return mB + np.log(wB.sum()) - mA - np.log(wA.sum()) - logT, (b[c + 1:] * wB).sum() / wB.sum() - (b[:c + 1] * wA).sum() / wA.sum()


def solve(a, b, c, logT, fix):
    ev = 1
    h0, dh = h(0, a, b, c, logT)
    if not h0 < 0:
        return 0, ev
    lo, hi = 0, -h0 / dh
    hh, _ = h(hi, a, b, c, logT); ev += 1
    while hh < 0:
        lo, hi = hi, 2 * hi
        hh, _ = h(hi, a, b, c, logT); ev += 1
    q = hi
    for it in range(100):
        hv, d = h(q, a, b, c, logT); ev += 1
        if hv < 0: lo = q
        else: hi = q
        step = hv / d
        if fix and abs(step) <= 4e-16 * abs(q): break
        qn = q - step
        if not (qn > lo and qn < hi): qn = .5 * (lo + hi)
        if abs(qn - q) <= 4e-16 * abs(qn) or hi - lo <= 4e-16 * hi:
            q = qn; break
        q = qn
    return q, ev


for name, pi, varg in (("cold start", [0.95, 0.02, 0.02, 0.01], 6e-4), ("converged", [0.992, 0.004, 0.003, 0.001], 1e-3)):
    new, old, mx = [], [], 0
    for t in range(3000):
        a, b = params(50000 * rng.uniform(0.1, 1.2), np.array(pi), varg, 0.5)
        U = rng.uniform(); logT = np.log((1 - U) / U)
        for c in range(3):
            q, ev = solve(a, b, c, logT, True); q0, ev0 = solve(a, b, c, logT, False)
            new.append(ev); old.append(ev0)
            if q0 > 0: mx = max(mx, abs(q - q0) / q0)
    for lab, t in (("before", np.array(old)), ("after", np.array(new))):
        print("%s, %s: evaluations per boundary mean %.1f p50 %d p90 %d p99 %d max %d" % (name, lab, t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max()))
    print("%s: largest relative change of a root %.2e" % (name, mx))
