"""GWAS windows — mirror of cutwind_by_bp / cutwind_by_num (reference src/cutwind.cpp:14-65).
Returns 1-based window ids per marker (what Bayes() takes as `windindx`, src/Bayes.cpp:84)."""
import numpy as np


def cutwind_by_bp(chrom, pos, bp):
    chrom = np.asarray(chrom)
    pos = np.asarray(pos, dtype=np.float64)
    wind = np.zeros(chrom.size, dtype=np.uint32)
    count = 1
    for c in np.unique(chrom):  # arma::unique sorts
        idx1 = np.flatnonzero(chrom == c)
        p = pos[idx1]
        bp0, maxbp = 1.0, p.max()
        while bp0 <= maxbp:
            sel = (p >= bp0) & (p < bp0 + bp)
            if sel.any():
                wind[idx1[sel]] = count
                count += 1
            bp0 += bp
    return wind


def cutwind_by_num(chrom, pos, fixN):
    chrom = np.asarray(chrom)
    pos = np.asarray(pos, dtype=np.float64)
    wind = np.zeros(chrom.size, dtype=np.uint32)
    count = 1
    for c in np.unique(chrom):
        idx1 = np.flatnonzero(chrom == c)
        L = idx1.size
        if L <= fixN:
            wind[idx1] = count
            count += 1
        else:
            order = np.argsort(pos[idx1], kind="stable")
            st = end = 0
            while end < L - 1:
                end = min(st + fixN - 1, L - 1)
                wind[idx1[order[st:end + 1]]] = count
                st += fixN
                count += 1
    return wind
