"""PLINK binary loader — mirror of read_plink() (reference R/read_plink.r:24-77) and of the
.bed decode it calls (reference src/read_bed.cpp:98-232).

Genotypes come back as an n x m int8 matrix in column-major order: the layout of the bigmemory
"char" matrix the reference fills (R/read_plink.r:57-65) and the one the device holds, so it can
be handed to `Bayes`/`ibrm` (or uploaded with hb_ctx_upload_genotype_i8) without a copy.
A1A1 -> 2, A1A2 -> 1, A2A2 -> 0 (mode "A"); mode "D" codes heterozygotes 1, homozygotes 0.
The device-side decoder (hb_ctx_upload_bed, k_bed_decode) applies the same map straight from the
2-bit file image; this module is the host loader for the file formats around it.
"""
import os

import numpy as np

NA_CHAR = -128


def read_bim(path):
    """rMap_c, reference src/read_bed.cpp:28-95: SNP, Chr, Pos, A1, A2."""
    snp, chrom, pos, a1, a2 = [], [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 6:
                continue
            chrom.append(p[0]); snp.append(p[1]); pos.append(float(p[3])); a1.append(p[4]); a2.append(p[5])
    return {"SNP": snp, "Chr": chrom, "Pos": np.array(pos), "A1": a1, "A2": a2}


def read_fam(path):
    with open(path) as f:
        return [line.split() for line in f if line.strip()]


def decode_bed(raw, nind, nsnp, impute=True, mode="A"):
    """2-bit SNP-major image -> int8 (nind x nsnp, Fortran order).
    Code map of reference src/read_bed.cpp:116-120: 00 -> 2, 01 -> NA, 10 -> 1, 11 -> 0
    (dominance mode: 0, NA, 1, 0); major-genotype imputation as :182-230."""
    raw = np.frombuffer(raw, dtype=np.uint8)
    bpc = (nind + 3) // 4
    if raw.size < 3 + bpc * nsnp:
        raise ValueError("bed file too short")
    if tuple(raw[:3]) != (0x6C, 0x1B, 0x01):
        raise ValueError("not a SNP-major PLINK .bed file")
    d = mode == "D"
    lut = np.array([0 if d else 2, NA_CHAR, 1, 0], dtype=np.int8)
    body = raw[3:3 + bpc * nsnp].reshape(nsnp, bpc)
    out = np.empty((nind, nsnp), dtype=np.int8, order="F")
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    for j0 in range(0, nsnp, 4096):
        blk = body[j0:j0 + 4096]
        codes = (blk[:, :, None] >> shifts[None, None, :]) & 3
        g = lut[codes.reshape(blk.shape[0], bpc * 4)[:, :nind]]
        if impute:
            miss = g == NA_CHAR
            for r in np.flatnonzero(miss.any(axis=1)):
                col = g[r]
                cand = (0, 1) if d else (0, 1, 2)
                counts = [int((col == v).sum()) for v in cand]
                best, major = 0, 0
                for v, cnt in zip(cand, counts):
                    if cnt > best:
                        best, major = cnt, v
                col[miss[r]] = major
        out[:, j0:j0 + blk.shape[0]] = g.T
    return out


def read_plink(bfile, maxLine=10000, impute=True, mode="A", out=None, threads=4):
    """Returns {"fam": rows of the .fam file, "geno": int8 n x m (F order), "map": bim columns}.
    When `out` is given the same side files as the reference are written: out.bin (the raw
    column-major int8 matrix, i.e. the bigmemory backing file), out.id and out.map."""
    if mode not in ("A", "D"):
        raise ValueError("'arg' should be one of 'A', 'D'")
    bim = read_bim(bfile + ".bim")
    fam = read_fam(bfile + ".fam")
    n, m = len(fam), len(bim["SNP"])
    bed = bfile if bfile.endswith(".bed") else bfile + ".bed"
    with open(bed, "rb") as f:
        raw = f.read()
    geno = decode_bed(raw, n, m, impute=impute, mode=mode)
    if out is not None:
        geno.T.tofile(out + ".bin")  # column-major bytes
        with open(out + ".id", "w") as f:
            f.write("\n".join(r[1] for r in fam) + "\n")
        with open(out + ".map", "w") as f:
            f.write("SNP\tChr\tPos\tA1\tA2\n")
            for i in range(m):
                f.write("%s\t%s\t%g\t%s\t%s\n" % (bim["SNP"][i], bim["Chr"][i], bim["Pos"][i], bim["A1"][i], bim["A2"][i]))
    return {"fam": fam, "geno": geno, "map": bim}


def read_table(path, sep="\t"):
    """Tiny stand-in for read.table(header=TRUE): dict of string columns, 'NA' kept as None."""
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f if l.strip()]
    hdr = lines[0].split(sep)
    cols = {h: [] for h in hdr}
    for l in lines[1:]:
        parts = l.split(sep)
        for h, v in zip(hdr, parts):
            cols[h].append(None if v in ("NA", "") else v)
    return cols
