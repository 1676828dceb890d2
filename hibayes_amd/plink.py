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
            chrom.append(p[0]); snp.append(p[1]); pos.append(p[3]); a1.append(p[4]); a2.append(p[5])
    # the reference keeps every column as text (vector<string>); Pos is also offered as numbers (NaN where it is not one)
    def num(v):
        try:
            return float(v)
        except ValueError:
            return float("nan")
    return {"SNP": snp, "Chr": chrom, "Pos": np.array([num(v) for v in pos]), "Pos_text": pos, "A1": a1, "A2": a2}


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
        write_bigmatrix(out, geno)
        with open(out + ".id", "w") as f:                       # R/read_plink.r:75
            f.write("\n".join(r[1] for r in fam) + "\n")
        with open(out + ".map", "w") as f:                      # rMap_c, src/read_bed.cpp:78-85: the .bim text, untouched
            f.write("SNP\tCHROM\tPOS\tA1\tA2\n")
            for i in range(m):
                f.write("%s\t%s\t%s\t%s\t%s\n" % (bim["SNP"][i], bim["Chr"][i], bim["Pos_text"][i], bim["A1"][i], bim["A2"][i]))
    return {"fam": fam, "geno": geno, "map": bim}


# --------------------------------------------------------------------------------------------
# bigmemory file-backed matrices: the on-disk form read_plink() leaves behind (R/read_plink.r:57-65) and that a later
# session re-opens with attach.big.matrix("xx.desc"). The .bin is the raw column-major array (type "char" = int8, NA =
# -128); the .desc is R's dput() of a big.matrix.descriptor. Third-party format: bigmemory (CRAN; hibayes DESCRIPTION
# Imports it without a version pin), files R/bigmemory.R (describe / attach.resource) and src/BigMatrix.cpp.
# --------------------------------------------------------------------------------------------
_BM_TYPES = {"char": np.int8, "short": np.int16, "integer": np.int32, "float": np.float32, "double": np.float64,
             "raw": np.uint8}


def write_bigmatrix(out, geno):
    """Write `geno` (n x m int8) as out.bin + out.desc, the pair bigmemory::big.matrix(backingfile=, descriptorfile=)
    creates for type = "char"."""
    g = np.asarray(geno)
    if g.dtype != np.int8:
        raise ValueError("the bigmemory 'char' matrix holds int8 genotype codes")
    n, m = g.shape
    np.asfortranarray(g).T.tofile(out + ".bin")   # column-major bytes
    d = os.path.dirname(os.path.abspath(out))
    with open(out + ".desc", "w") as f:
        f.write('new("big.matrix.descriptor", description = list(sharedType = "FileBacked", \n'
                '    filename = "%s", dirname = "%s/", totalRows = %dL, \n'
                '    totalCols = %dL, rowOffset = c(0, %d), colOffset = c(0, \n'
                '    %d), nrow = %d, ncol = %d, rowNames = NULL, colNames = NULL, \n'
                '    type = "char", separated = FALSE))\n' % (os.path.basename(out) + ".bin", d, n, m, n, m, n, m))


def parse_bigmatrix_desc(text):
    """Fields of a big.matrix.descriptor as dput() prints it (whitespace and line breaks anywhere)."""
    import re
    t = " ".join(text.split())
    if "big.matrix.descriptor" not in t:
        raise ValueError("not a bigmemory descriptor file")

    def field(name, pat):
        mm = re.search(name + r"\s*=\s*" + pat, t)
        return mm.group(1) if mm else None

    def pair(name):
        mm = re.search(name + r"\s*=\s*c\(\s*([-0-9.eE+]+)L?\s*,\s*([-0-9.eE+]+)L?\s*\)", t)
        return (int(float(mm.group(1))), int(float(mm.group(2)))) if mm else None

    d = {"filename": field("filename", r'"([^"]*)"'), "dirname": field("dirname", r'"([^"]*)"'),
         "type": field("type", r'"([^"]*)"'), "sharedType": field("sharedType", r'"([^"]*)"'),
         "separated": field("separated", r"(TRUE|FALSE)") == "TRUE"}
    for k in ("totalRows", "totalCols", "nrow", "ncol"):
        v = field(k, r"([-0-9.eE+]+)L?")
        d[k] = None if v is None else int(float(v))
    d["rowOffset"], d["colOffset"] = pair("rowOffset"), pair("colOffset")
    if d["filename"] is None or d["totalRows"] is None or d["totalCols"] is None or d["type"] is None:
        raise ValueError("incomplete bigmemory descriptor")
    return d


def attach_bigmatrix(descfile, backingpath=None):
    """attach.big.matrix(descfile): memory-map the backing file and return the n x m matrix WITHOUT loading it
    (numpy.memmap, Fortran order) — for type "char" this is exactly the int8 column-major layout hb_ctx_upload_genotype_i8
    / Bayes(X_i8) take, so a 25 GB genotype file goes from disk to the device without a host copy of doubles
    (the reference's as.matrix() at R/bayes.r:284 makes one of 8 bytes per genotype). A sub-matrix descriptor
    (rowOffset / colOffset) is honoured. `backingpath` overrides the recorded directory (files that were moved)."""
    with open(descfile) as f:
        d = parse_bigmatrix_desc(f.read())
    if d["separated"]:
        raise ValueError("separated (one file per column) big.matrix descriptors are not supported")
    if d["type"] not in _BM_TYPES:
        raise ValueError("unknown big.matrix type '%s'" % d["type"])
    cands = []
    if backingpath is not None:
        cands.append(os.path.join(backingpath, d["filename"]))
    cands.append(os.path.join(os.path.dirname(os.path.abspath(descfile)), d["filename"]))
    if d["dirname"]:
        cands.append(os.path.join(d["dirname"], d["filename"]))
    path = next((c for c in cands if os.path.exists(c)), None)
    if path is None:
        raise FileNotFoundError("backing file '%s' of %s not found" % (d["filename"], descfile))
    dt = np.dtype(_BM_TYPES[d["type"]])
    R, C = d["totalRows"], d["totalCols"]
    if os.path.getsize(path) < R * C * dt.itemsize:
        raise ValueError("backing file shorter than totalRows x totalCols")
    mm = np.memmap(path, dtype=dt, mode="r", shape=(R, C), order="F")
    r0, nr = d["rowOffset"] if d["rowOffset"] else (0, R)
    c0, nc = d["colOffset"] if d["colOffset"] else (0, C)
    return mm[r0:r0 + nr, c0:c0 + nc]


def read_bigmatrix(prefix):
    """The four files read_plink(out = prefix) leaves: genotypes (memory-mapped), individual ids, map."""
    geno = attach_bigmatrix(prefix + ".desc")
    ids = None
    if os.path.exists(prefix + ".id"):
        with open(prefix + ".id") as f:
            ids = [l.strip() for l in f if l.strip()]
    mp = None
    if os.path.exists(prefix + ".map"):
        t = read_table(prefix + ".map")
        mp = {"SNP": t["SNP"], "Chr": t["CHROM"], "Pos_text": t["POS"],
              "Pos": np.array([float(v) if v is not None else float("nan") for v in t["POS"]]), "A1": t["A1"], "A2": t["A2"]}
    return {"geno": geno, "id": ids, "map": mp}


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
