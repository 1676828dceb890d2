/*
 * examples/ibrm_demo.c — a caller on the C side of the boundary: the ibrm(T1 ~ 1, BayesCpi) example of the reference
 * (R/bayes.r:80-97) in plain C99 against include/hibayes_gpu.h, nothing else.
 *
 *     gcc -std=c99 -O2 -Iinclude examples/ibrm_demo.c -Lhibayes_amd -lhibayes_gpu -Wl,-rpath,$PWD/hibayes_amd -o ibrm_demo
 *     ./ibrm_demo tests/golden/demo/demo [niter nburn thin [bits]]
 *
 * What ibrm() does before it calls Bayes() is done here the same way: individuals = rows of the .fam file that have a
 * phenotype record that is not NA (R/bayes.r:161-165, :199-207), y = their T1, the genotypes of exactly those rows
 * (R/bayes.r:286-291). The .bed image goes to the device as it is and is decoded there (hb_ctx_upload_bed: code map and
 * major-genotype imputation of src/read_bed.cpp:116-120, :182-230), then hb_bayes_run() takes the pre-loaded context.
 * Prints the fields of the result list (src/Bayes.cpp:919-1040) one per line, "name value ...", 17 significant digits.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hibayes_gpu.h"

static void die(const char *what)
{
    fprintf(stderr, "ibrm_demo: %s: %s\n", what, hb_last_error());
    exit(1);
}

static char *slurp(const char *path, long *size)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ibrm_demo: cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END);
    *size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)*size + 1);
    if (fread(buf, 1, (size_t)*size, f) != (size_t)*size) { fprintf(stderr, "ibrm_demo: short read of %s\n", path); exit(1); }
    buf[*size] = 0;
    fclose(f);
    return buf;
}

/* field `k` (0-based, separated by blanks or tabs) of a line, copied into out */
static int field(const char *line, int k, char *out, size_t cap)
{
    const char *p = line;
    for (int i = 0;; i++) {
        while (*p == ' ' || *p == '\t') p++;
        if (!*p || *p == '\n' || *p == '\r') return 0;
        const char *q = p;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;
        if (i == k) {
            size_t len = (size_t)(q - p) < cap - 1 ? (size_t)(q - p) : cap - 1;
            memcpy(out, p, len);
            out[len] = 0;
            return 1;
        }
        p = q;
    }
}

static int count_lines(const char *txt)
{
    int n = 0;
    for (const char *p = txt; *p; p++) n += *p == '\n';
    return n;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ibrm_demo <plink prefix> [niter nburn thin [bits]]\n"); return 2; }
    const int niter = argc > 2 ? atoi(argv[2]) : 2000, nburn = argc > 3 ? atoi(argv[3]) : 1200, thin = argc > 4 ? atoi(argv[4]) : 5;
    const int bits = argc > 5 ? atoi(argv[5]) : 8;
    char path[4096];
    long sz;

    /* .fam: one genotyped individual per line, id in column 2; .bim: one marker per line */
    snprintf(path, sizeof path, "%s.fam", argv[1]);
    char *fam = slurp(path, &sz);
    const int nind = count_lines(fam);
    snprintf(path, sizeof path, "%s.bim", argv[1]);
    char *bim = slurp(path, &sz);
    const int m = count_lines(bim);
    char(*ids)[64] = malloc((size_t)nind * 64);
    {
        const char *p = fam;
        for (int i = 0; i < nind; i++) {
            field(p, 1, ids[i], 64);
            p = strchr(p, '\n') + 1;
        }
    }
    /* .phe: header line, id in the first column, the trait T1 in the column the header names */
    snprintf(path, sizeof path, "%s.phe", argv[1]);
    char *phe = slurp(path, &sz);
    int tcol = -1;
    {
        char name[64];
        for (int k = 0; field(phe, k, name, sizeof name); k++)
            if (!strcmp(name, "T1")) tcol = k;
    }
    if (tcol < 0) { fprintf(stderr, "ibrm_demo: no column T1 in %s\n", path); return 1; }
    /* rows of the .fam (in its order) with a non-missing phenotype: the individuals of the fit */
    int32_t *rows = malloc(sizeof(int32_t) * (size_t)nind);
    double *y = malloc(sizeof(double) * (size_t)nind);
    int n = 0;
    for (int i = 0; i < nind; i++) {
        for (const char *p = strchr(phe, '\n') + 1; p && *p; p = strchr(p, '\n') ? strchr(p, '\n') + 1 : NULL) {
            char id[64], val[64];
            if (!field(p, 0, id, sizeof id) || strcmp(id, ids[i])) continue;
            if (field(p, tcol, val, sizeof val) && strcmp(val, "NA")) {
                rows[n] = i;
                y[n++] = atof(val);
            }
            break;
        }
    }
    snprintf(path, sizeof path, "%s.bed", argv[1]);
    char *bed = slurp(path, &sz);

    if (hb_abi_version() != HB_ABI_VERSION) { fprintf(stderr, "ibrm_demo: header / library ABI mismatch\n"); return 1; }
    hb_ctx_params cp;
    memset(&cp, 0, sizeof cp);
    cp.n = n;
    cp.m = m;
    cp.precise = 2;
    cp.seed = 666666;
    hb_ctx *ctx = NULL;
    if (hb_ctx_create(&cp, &ctx)) die("hb_ctx_create");
    if (hb_ctx_upload_bed(ctx, (const uint8_t *)bed, sz, nind, rows, 0, m)) die("hb_ctx_upload_bed");

    static const double Pi[2] = {0.95, 0.05};
    hb_bayes_args a;
    memset(&a, 0, sizeof a);
    a.n = n;
    a.m = m;
    a.y = y;
    a.model = "BayesCpi";
    a.Pi = Pi;
    a.n_pi = 2;
    a.niter = niter;
    a.nburn = nburn;
    a.thin = thin;
    a.outfreq = 100;
    a.seed = 666666; /* ibrm()'s default seed (R/bayes.r:148) */
    a.precise = 2;
    a.ctx = ctx;
    if (bits == 2 && hb_ctx_set_layout(ctx, 2, 0)) die("hb_ctx_set_layout"); /* (the Gram blocks are rebuilt from an unpacked copy when the run asks for them) */

    hb_bayes_out o;
    memset(&o, 0, sizeof o);
    double pi[2], *alpha = malloc(sizeof(double) * (size_t)m), *pip = malloc(sizeof(double) * (size_t)m);
    double *g = malloc(sizeof(double) * (size_t)n), *e = malloc(sizeof(double) * (size_t)n);
    o.pi = pi;
    o.alpha = alpha;
    o.pip = pip;
    o.g = g;
    o.e = e;
    if (hb_bayes_run(&a, &o)) die("hb_bayes_run");

    printf("n %d\nm %d\nn_records %d\nnzct %d\n", n, m, o.n_records, o.nzct);
    printf("Vg %.17g\nVe %.17g\nh2 %.17g\nmu %.17g\npi %.17g %.17g\n", o.Vg, o.Ve, o.h2, o.mu, pi[0], pi[1]);
    printf("alpha");
    for (int j = 0; j < m; j++) printf(" %.17g", alpha[j]);
    printf("\npip");
    for (int j = 0; j < m; j++) printf(" %.17g", pip[j]);
    printf("\n");
    hb_ctx_destroy(ctx);
    return 0;
}
