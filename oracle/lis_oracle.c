/*
 * lis_oracle.c -- CPU oracle (checker) for the Lis SpMV + Krylov hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see lis_oracle.h.  The product (lis_amd/) never links this file.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA: the reference's default x86-64 build has none, so each
 * `t += a*b` is one rounded multiply followed by one rounded add).
 *
 * Status: PINNED.  tests/test_oracle_vs_ref.py checks every function below bit-for-bit against
 * oracle/_ref/liblis_ref.so (the reference compiled from /root/reference/src, OMP threads = 1) and
 * tests/test_golden.py checks it against the committed fixtures under tests/golden/.
 */
#include "lis_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================== synthetic inputs ============================== */

int orc_gen_poisson1d(int gn, int is, int ie, int *ptr, int *idx, double *val)
{
    int c = 0;
    ptr[0] = 0;
    for (int g = is; g < ie; g++) {
        if (g > 0)      { idx[c] = g - 1; val[c++] = -1.0; }
        if (g < gn - 1) { idx[c] = g + 1; val[c++] = -1.0; }
        idx[c] = g; val[c++] = 2.0;
        ptr[g - is + 1] = c;
    }
    return c;
}

static void row_insertion_sort(int lo, int hi, int *idx, double *val)
{
    /* rows of the stencil have distinct columns, so any sort gives the reference's order */
    for (int a = lo + 1; a < hi; a++) {
        int ci = idx[a]; double cv = val[a]; int b = a - 1;
        while (b >= lo && idx[b] > ci) { idx[b + 1] = idx[b]; val[b + 1] = val[b]; b--; }
        idx[b + 1] = ci; val[b + 1] = cv;
    }
}

int orc_gen_poisson3d(int l, int m, int n, int is, int ie, int sort_cols,
                      int *ptr, int *idx, double *val)
{
    const int plane = m * n;
    int c = 0;
    ptr[0] = 0;
    for (int g = is; g < ie; g++) {
        int i = g / plane, rem = g - i * plane, j = rem / n, k = rem - j * n;
        int start = c;
        if (i > 0)     { idx[c] = g - plane; val[c++] = -1.0; }
        if (i < l - 1) { idx[c] = g + plane; val[c++] = -1.0; }
        if (j > 0)     { idx[c] = g - n;     val[c++] = -1.0; }
        if (j < m - 1) { idx[c] = g + n;     val[c++] = -1.0; }
        if (k > 0)     { idx[c] = g - 1;     val[c++] = -1.0; }
        if (k < n - 1) { idx[c] = g + 1;     val[c++] = -1.0; }
        idx[c] = g; val[c++] = 6.0;
        if (sort_cols) row_insertion_sort(start, c, idx, val);
        ptr[g - is + 1] = c;
    }
    return c;
}

/* ================================== SpMV ================================== */

void orc_spmv_csr(int n, const int *ptr, const int *idx, const double *val,
                  const double *x, double *y)
{
    for (int r = 0; r < n; r++) {
        double acc = 0.0;
        for (int k = ptr[r]; k < ptr[r + 1]; k++) acc += val[k] * x[idx[k]];
        y[r] = acc;
    }
}

void orc_spmv_csc(int n, int np, const int *ptr, const int *idx, const double *val,
                  const double *x, double *y)
{
    for (int r = 0; r < n; r++) y[r] = 0.0;
    for (int c = 0; c < np; c++) {
        const double xc = x[c];
        for (int k = ptr[c]; k < ptr[c + 1]; k++) y[idx[k]] += val[k] * xc;
    }
}

void orc_spmv_ell(int n, int maxnzr, const int *idx, const double *val,
                  const double *x, double *y)
{
    for (int r = 0; r < n; r++) y[r] = 0.0;
    for (int j = 0; j < maxnzr; j++) {
        const size_t base = (size_t)j * (size_t)n;
        for (int r = 0; r < n; r++) y[r] += val[base + r] * x[idx[base + r]];
    }
}

/* ---- y = A^T x (lis_matvech_<fmt>), the reference's scatter loops at one thread.  The OpenMP build scatters
 * into a zeroed per-thread buffer w and stores y[i] = 0.0 + w[i]; with one thread that is the same value as the
 * sequential sums below (w starts +0.0 and can never become -0.0). */
void orc_spmvh_csr(int n, int np, const int *ptr, const int *idx, const double *val,
                   const double *x, double *y)
{   /* lis_matvec_csr.c:213-250 */
    for (int c = 0; c < np; c++) y[c] = 0.0;
    for (int r = 0; r < n; r++) {
        const double t = x[r];
        for (int k = ptr[r]; k < ptr[r + 1]; k++) y[idx[k]] += val[k] * t;
    }
}

void orc_spmvh_csc(int np, const int *ptr, const int *idx, const double *val,
                   const double *x, double *y)
{   /* lis_matvec_csc.c:176-190: row sums over the stored columns */
    for (int c = 0; c < np; c++) {
        double t = 0.0;
        for (int k = ptr[c]; k < ptr[c + 1]; k++) t += val[k] * x[idx[k]];
        y[c] = t;
    }
}

void orc_spmvh_ell(int n, int np, int maxnzr, const int *idx, const double *val,
                   const double *x, double *y)
{   /* lis_matvec_ell.c:181-215: jagged column by jagged column, padding entries included */
    for (int c = 0; c < np; c++) y[c] = 0.0;
    for (int j = 0; j < maxnzr; j++) {
        const size_t base = (size_t)j * (size_t)n;
        for (int r = 0; r < n; r++) y[idx[base + r]] += val[base + r] * x[r];
    }
}

void orc_spmvh_dia(int n, int np, int nnd, const int *off, const double *val,
                   const double *x, double *y)
{   /* lis_matvec_dia.c:262-310 with one chunk: diagonal by diagonal */
    for (int c = 0; c < np; c++) y[c] = 0.0;
    for (int d = 0; d < nnd; d++) {
        const int jj = off[d];
        const int lo = jj < 0 ? -jj : 0, hi = (n - jj) < n ? (n - jj) : n;
        for (int r = lo; r < hi; r++) y[jj + r] += val[(size_t)d * (size_t)n + r] * x[r];
    }
}

void orc_spmvh_jad(int n, int np, int maxnzr, const int *perm, const int *ptr, const int *idx,
                   const double *val, const double *x, double *y)
{   /* lis_matvec_jad.c:545-580 */
    (void)n;
    for (int c = 0; c < np; c++) y[c] = 0.0;
    for (int j = 0; j < maxnzr; j++) {
        int k = 0;
        for (int i = ptr[j]; i < ptr[j + 1]; i++, k++) y[idx[i]] += val[i] * x[perm[k]];
    }
}

void orc_spmvh_bsr(int nr, int bnr, int bnc, const int *bptr, const int *bidx, const double *val,
                   const double *x, double *y, int ylen)
{   /* lis_matvec_bsr.c:935-957: x and y are the padded vectors (nr*bnr / nc*bnc entries) */
    for (int c = 0; c < ylen; c++) y[c] = 0.0;
    const int bs = bnr * bnc;
    for (int bi = 0; bi < nr; bi++)
        for (int bc = bptr[bi]; bc < bptr[bi + 1]; bc++) {
            const int bj = bidx[bc] * bnc;
            size_t k = (size_t)bc * bs;
            for (int j = 0; j < bnc; j++)
                for (int i = 0; i < bnr; i++, k++) y[bj + j] += val[k] * x[bi * bnr + i];
        }
}

static void chunk_range(int id, int nchunks, int n, int *lo, int *hi)
{   /* static row split used by every reference kernel: LIS_GET_ISIE, include/lis.h:1067-1078 */
    int q = n / nchunks, rem = n % nchunks;
    if (id < rem) { *lo = (q + 1) * id;     *hi = *lo + q + 1; }
    else          { *lo = q * id + rem;     *hi = *lo + q;     }
}

void orc_spmv_dia(int n, int nnd, int nchunks, const int *off, const double *val,
                  const double *x, double *y)
{
    for (int c = 0; c < nchunks; c++) {
        int lo, hi; chunk_range(c, nchunks, n, &lo, &hi);
        const int len = hi - lo;
        for (int r = lo; r < hi; r++) y[r] = 0.0;
        for (int d = 0; d < nnd; d++) {
            const int o = off[d];
            int rs = lo > -o ? lo : -o;
            int re = hi < n - o ? hi : n - o;
            const double *v = val + (size_t)lo * nnd + (size_t)d * len;
            for (int r = rs; r < re; r++) y[r] += v[r - lo] * x[o + r];
        }
    }
}

void orc_spmv_jad(int n, int maxnzr, int nchunks, const int *perm, const int *ptr,
                  const int *idx, const double *val, const double *x, double *y)
{
    double *w = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    for (int c = 0; c < nchunks; c++) {
        int lo, hi; chunk_range(c, nchunks, n, &lo, &hi);
        const int *p = ptr + (size_t)c * (maxnzr + 1);
        for (int r = lo; r < hi; r++) w[r] = 0.0;
        for (int j = 0; j < maxnzr; j++) {
            int slot = lo;
            for (int k = p[j]; k < p[j + 1]; k++) { w[slot] += val[k] * x[idx[k]]; slot++; }
        }
        for (int r = lo; r < hi; r++) y[perm[r]] = w[r];
    }
    free(w);
}

void orc_spmv_bsr(int n, int nr, int bnr, int bnc, const int *bptr, const int *bidx,
                  const double *val, const double *x, double *y)
{
    const int bs = bnr * bnc;
    (void)n;
    for (int br = 0; br < nr; br++) {
        double *yr = y + (size_t)br * bnr;
        for (int i = 0; i < bnr; i++) yr[i] = 0.0;
        for (int b = bptr[br]; b < bptr[br + 1]; b++) {
            const double *blk = val + (size_t)b * bs;       /* column-major bnr x bnc */
            const double *xb  = x + (size_t)bidx[b] * bnc;
            for (int j = 0; j < bnc; j++)
                for (int i = 0; i < bnr; i++) yr[i] += blk[j * bnr + i] * xb[j];
        }
    }
}

/* ============================ format conversions ============================ */

int orc_ell_maxnzr(int n, const int *ptr)
{
    int mx = 0;
    for (int r = 0; r < n; r++) { int c = ptr[r + 1] - ptr[r]; if (c > mx) mx = c; }
    return mx;
}
int orc_jad_maxnzr(int n, const int *ptr) { return orc_ell_maxnzr(n, ptr); }

void orc_csr2ell(int n, const int *ptr, const int *idx, const double *val,
                 int maxnzr, int *eidx, double *eval)
{
    for (int j = 0; j < maxnzr; j++)
        for (int r = 0; r < n; r++) { eval[(size_t)j * n + r] = 0.0; eidx[(size_t)j * n + r] = r; }
    for (int r = 0; r < n; r++) {
        int j = 0;
        for (int k = ptr[r]; k < ptr[r + 1]; k++, j++) {
            eval[(size_t)j * n + r] = val[k];
            eidx[(size_t)j * n + r] = idx[k];
        }
    }
}

void orc_csr2csc(int n, int np, const int *ptr, const int *idx, const double *val,
                 int *cptr, int *cidx, double *cval)
{
    int *fill = (int *)calloc((size_t)np + 1, sizeof(int));
    for (int r = 0; r < n; r++)
        for (int k = ptr[r]; k < ptr[r + 1]; k++) fill[idx[k]]++;
    cptr[0] = 0;
    for (int c = 0; c < np; c++) { cptr[c + 1] = cptr[c] + fill[c]; fill[c] = cptr[c]; }
    for (int r = 0; r < n; r++)
        for (int k = ptr[r]; k < ptr[r + 1]; k++) {
            int dst = fill[idx[k]]++;
            cval[dst] = val[k];
            cidx[dst] = r;
        }
    free(fill);
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

int orc_csr2dia(int n, int nnz, const int *ptr, const int *idx, const double *val,
                int *off, double *dval)
{
    if (nnz <= 0) return 0;
    int *d = (int *)malloc((size_t)nnz * sizeof(int));
    for (int r = 0; r < n; r++)
        for (int k = ptr[r]; k < ptr[r + 1]; k++) d[k] = idx[k] - r;
    qsort(d, (size_t)nnz, sizeof(int), cmp_int);      /* a set of ints: any sort gives the same list */
    int nnd = 1;
    for (int k = 1; k < nnz; k++) if (d[k] != d[k - 1]) nnd++;
    if (off) {
        int w = 0;
        off[w++] = d[0];
        for (int k = 1; k < nnz; k++) if (d[k] != d[k - 1]) off[w++] = d[k];
        memset(dval, 0, (size_t)n * nnd * sizeof(double));
        for (int r = 0; r < n; r++) {
            int slot = 0;                                   /* rows are column-sorted: slot only advances */
            for (int k = ptr[r]; k < ptr[r + 1]; k++) {
                int o = idx[k] - r;
                while (off[slot] != o) slot++;
                dval[(size_t)slot * n + r] = val[k];
            }
        }
    }
    free(d);
    return nnd;
}

/* descending sort of key[] carrying tag[]: the reference's own scheme (middle pivot parked at the end,
 * Hoare scan with strict comparisons) restated, because the tie order among equal-length rows -- and
 * with it the JAD permutation -- depends on it.  src/system/lis_sort.c:249-276. */
static void sort_desc_pairs(int lo, int hi, int *key, int *tag)
{
    while (lo < hi) {
        int mid = (lo + hi) / 2, pv = key[mid], t;
        t = key[mid]; key[mid] = key[hi]; key[hi] = t;
        t = tag[mid]; tag[mid] = tag[hi]; tag[hi] = t;
        int a = lo, b = hi;
        while (a <= b) {
            while (key[a] > pv) a++;
            while (key[b] < pv) b--;
            if (a <= b) {
                t = key[a]; key[a] = key[b]; key[b] = t;
                t = tag[a]; tag[a] = tag[b]; tag[b] = t;
                a++; b--;
            }
        }
        sort_desc_pairs(lo, b, key, tag);   /* left part by recursion, right part by iteration */
        lo = a;
    }
}

void orc_csr2jad(int n, const int *ptr, const int *idx, const double *val, int maxnzr,
                 int *perm, int *jptr, int *jidx, double *jval)
{
    int *len = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    memset(jptr, 0, (size_t)(maxnzr + 1) * sizeof(int));
    for (int r = 0; r < n; r++) {
        len[r] = ptr[r + 1] - ptr[r];
        perm[r] = r;
        for (int j = 0; j < len[r]; j++) jptr[j + 1]++;
    }
    sort_desc_pairs(0, n - 1, len, perm);
    jptr[0] = 0;
    for (int j = 0; j < maxnzr; j++) jptr[j + 1] += jptr[j];
    for (int s = 0; s < n; s++) {
        int src = ptr[perm[s]], cnt = ptr[perm[s] + 1] - src;
        for (int j = 0; j < cnt; j++) {
            int dst = jptr[j] + s;
            jval[dst] = val[src + j];
            jidx[dst] = idx[src + j];
        }
    }
    free(len);
}

int orc_csr2bsr(int n, const int *ptr, const int *idx, const double *val, int bnr, int bnc,
                int *bptr, int *bidx, double *bval)
{
    const int nr = 1 + (n - 1) / bnr, nc = 1 + (n - 1) / bnc, bs = bnr * bnc;
    int *slot = (int *)calloc((size_t)nc, sizeof(int));     /* 0 = block column not yet seen in this block row */
    int *seen = (int *)malloc(sizeof(int) * (size_t)nc);
    bptr[0] = 0;
    for (int br = 0; br < nr; br++) {                        /* pass 1: count distinct block columns */
        int cnt = 0;
        int nseen = 0;
        for (int i = 0; i < bnr && br * bnr + i < n; i++)
            for (int k = ptr[br * bnr + i]; k < ptr[br * bnr + i + 1]; k++) {
                int bc = idx[k] / bnc;
                if (!slot[bc]) { slot[bc] = 1; seen[nseen++] = bc; cnt++; }
            }
        for (int s = 0; s < nseen; s++) slot[seen[s]] = 0;
        bptr[br + 1] = bptr[br] + cnt;
    }
    free(seen);
    const int bnnz = bptr[nr];
    if (bidx) {
        for (int br = 0; br < nr; br++) {                    /* pass 2: blocks in first-seen order */
            int next = bptr[br];
            for (int i = 0; i < bnr && br * bnr + i < n; i++)
                for (int k = ptr[br * bnr + i]; k < ptr[br * bnr + i + 1]; k++) {
                    int bc = idx[k] / bnc, jc = idx[k] % bnc;
                    if (!slot[bc]) {
                        slot[bc] = next * bs + 1;
                        bidx[next] = bc;
                        for (int z = 0; z < bs; z++) bval[(size_t)next * bs + z] = 0.0;
                        next++;
                    }
                    bval[(size_t)slot[bc] - 1 + (size_t)jc * bnr + i] = val[k];
                }
            for (int b = bptr[br]; b < bptr[br + 1]; b++) slot[bidx[b]] = 0;
        }
    }
    free(slot);
    return bnnz;
}

void orc_csr_diagonal(int n, const int *ptr, const int *idx, const double *val, double *d)
{
    for (int r = 0; r < n; r++) {
        d[r] = 0.0;
        for (int k = ptr[r]; k < ptr[r + 1]; k++) if (idx[k] == r) { d[r] = val[k]; break; }
    }
}

/* ================================ vector kernels ================================ */

double orc_dot(int n, const double *x, const double *y)
{
    double part = 0.0;
    for (int i = 0; i < n; i++) part += x[i] * y[i];
    double s = 0.0; s += part;      /* the reference folds its per-thread partials into a fresh 0.0 */
    return s;
}
double orc_nrm2(int n, const double *x)
{
    double part = 0.0;
    for (int i = 0; i < n; i++) part += x[i] * x[i];
    double s = 0.0; s += part;
    return sqrt(s);
}
double orc_nrm1(int n, const double *x)
{
    double part = 0.0;
    for (int i = 0; i < n; i++) part += fabs(x[i]);
    double s = 0.0; s += part;
    return s;
}
void orc_axpy (int n, double a, const double *x, double *y) { for (int i = 0; i < n; i++) y[i] += a * x[i]; }
void orc_xpay (int n, const double *x, double a, double *y) { for (int i = 0; i < n; i++) y[i] = x[i] + a * y[i]; }
void orc_axpyz(int n, double a, const double *x, const double *y, double *z) { for (int i = 0; i < n; i++) z[i] = a * x[i] + y[i]; }
void orc_scale(int n, double a, double *x) { for (int i = 0; i < n; i++) x[i] = a * x[i]; }
void orc_pmul (int n, const double *x, const double *y, double *z) { for (int i = 0; i < n; i++) z[i] = x[i] * y[i]; }
void orc_reciprocal(int n, double *x) { for (int i = 0; i < n; i++) x[i] = 1.0 / x[i]; }

/* ================================ Krylov loops ================================ */

typedef struct {
    int n; const int *ptr, *idx; const double *val;
    int precon; double *dinv;       /* Jacobi: 1/diag, lis_precon_jacobi.c:61-85 */
} orc_sys;

static void sys_open(orc_sys *S, int n, const int *ptr, const int *idx, const double *val, int precon)
{
    S->n = n; S->ptr = ptr; S->idx = idx; S->val = val; S->precon = precon; S->dinv = NULL;
    if (precon == ORC_PRECON_JACOBI) {
        S->dinv = (double *)malloc((size_t)n * sizeof(double));
        orc_csr_diagonal(n, ptr, idx, val, S->dinv);
        orc_reciprocal(n, S->dinv);
    }
}
static void sys_close(orc_sys *S) { free(S->dinv); }
static void sys_matvec(const orc_sys *S, const double *x, double *y)
{ orc_spmv_csr(S->n, S->ptr, S->idx, S->val, x, y); }
static void sys_psolve(const orc_sys *S, const double *b, double *x)
{   /* lis_precon.c:365-384 (none == copy), lis_precon_jacobi.c:121-124 */
    if (S->precon == ORC_PRECON_JACOBI) orc_pmul(S->n, b, S->dinv, x);
    else memcpy(x, b, (size_t)S->n * sizeof(double));
}
static double *vec_new(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* lis_solver.c:957-1091 for conv_cond nrm2_r: r = b (x0 = 0) or b - A x0; bnrm = 1/||r||; returns 1 when
 * already converged (the reference then reports iter = 1). */
static int initial_residual(const orc_sys *S, const double *b, const double *x, int init_zero,
                            double *r, double tol, double *bnrm, orc_result *out)
{
    const int n = S->n;
    if (!init_zero) { sys_matvec(S, x, r); orc_xpay(n, b, -1.0, r); }
    else memcpy(r, b, (size_t)n * sizeof(double));
    double nrm = orc_nrm2(n, r);
    *bnrm = (nrm == 0.0) ? 1.0 : 1.0 / nrm;
    nrm = nrm * *bnrm;
    if (nrm <= fabs(tol)) { out->retcode = 0; out->iter = 1; out->resid = nrm; return 1; }
    return 0;
}

orc_result orc_cg(int n, const int *ptr, const int *idx, const double *val,
                  const double *b, double *x, int precon, double tol, int maxiter,
                  int init_zero, double *rhistory)
{
    orc_result out = {0, 0, 0.0};
    orc_sys S; sys_open(&S, n, ptr, idx, val, precon);
    double *z = vec_new(n), *q = vec_new(n), *r = vec_new(n), *p = vec_new(n);
    double bnrm, nrm = 0.0, rho_old = 1.0;
    if (rhistory) rhistory[0] = 1.0;
    if (init_zero) memset(x, 0, (size_t)n * sizeof(double));
    if (initial_residual(&S, b, x, init_zero, r, tol, &bnrm, &out)) goto done;
    int it;
    for (it = 1; it <= maxiter; it++) {
        sys_psolve(&S, r, z);
        double rho = orc_dot(n, r, z);
        double beta = rho / rho_old;
        orc_xpay(n, z, beta, p);
        sys_matvec(&S, p, q);
        double pq = orc_dot(n, p, q);
        if (pq == 0.0) { out.retcode = 2; out.iter = it; out.resid = nrm; goto done; }
        double alpha = rho / pq;
        orc_axpy(n, alpha, p, x);
        orc_axpy(n, -alpha, q, r);
        nrm = orc_nrm2(n, r) * bnrm;
        if (rhistory) rhistory[it] = nrm;
        if (tol >= nrm) { out.retcode = 0; out.iter = it; out.resid = nrm; goto done; }
        rho_old = rho;
    }
    out.retcode = 4; out.iter = it; out.resid = nrm;
done:
    free(z); free(q); free(r); free(p); sys_close(&S);
    return out;
}

orc_result orc_bicgstab(int n, const int *ptr, const int *idx, const double *val,
                        const double *b, double *x, int precon, double tol, int maxiter,
                        int init_zero, double *rhistory)
{
    orc_result out = {0, 0, 0.0};
    orc_sys S; sys_open(&S, n, ptr, idx, val, precon);
    double *rt = vec_new(n), *r = vec_new(n), *t = vec_new(n), *p = vec_new(n),
           *v = vec_new(n), *ph = vec_new(n), *sh = vec_new(n);
    double *s = r;                                   /* s aliases r: lis_solver_bicgstab.c:160-161 */
    double bnrm, nrm = 0.0, alpha = 1.0, omega = 1.0, rho_old = 1.0;
    if (rhistory) rhistory[0] = 1.0;
    if (init_zero) memset(x, 0, (size_t)n * sizeof(double));
    if (initial_residual(&S, b, x, init_zero, r, tol, &bnrm, &out)) goto done;
    memcpy(rt, r, (size_t)n * sizeof(double));       /* shadow residual = r0, lis_solver.c:1862-1863 */
    int it;
    for (it = 1; it <= maxiter; it++) {
        double rho = orc_dot(n, rt, r);
        if (rho == 0.0) { out.retcode = 2; out.iter = it; out.resid = nrm; goto done; }
        if (it == 1) memcpy(p, r, (size_t)n * sizeof(double));
        else {
            double beta = (rho / rho_old) * (alpha / omega);
            orc_axpy(n, -omega, v, p);
            orc_xpay(n, r, beta, p);
        }
        sys_psolve(&S, p, ph);
        sys_matvec(&S, ph, v);
        double d1 = orc_dot(n, rt, v);
        alpha = rho / d1;
        orc_axpy(n, -alpha, v, r);                   /* s = r - alpha v (in place) */
        nrm = orc_nrm2(n, s) * bnrm;
        if (nrm <= tol) {
            if (rhistory) rhistory[it] = nrm;
            orc_axpy(n, alpha, ph, x);
            out.retcode = 0; out.iter = it; out.resid = nrm; goto done;
        }
        sys_psolve(&S, s, sh);
        sys_matvec(&S, sh, t);
        d1 = orc_dot(n, t, s);
        double d2 = orc_dot(n, t, t);
        omega = d1 / d2;
        orc_axpy(n, alpha, ph, x);
        orc_axpy(n, omega, sh, x);
        orc_axpy(n, -omega, t, r);
        nrm = orc_nrm2(n, r) * bnrm;
        if (rhistory) rhistory[it] = nrm;
        if (tol >= nrm) { out.retcode = 0; out.iter = it; out.resid = nrm; goto done; }
        if (omega == 0.0) { out.retcode = 2; out.iter = it; out.resid = nrm; goto done; }
        rho_old = rho;
    }
    out.retcode = 4; out.iter = it; out.resid = nrm;
done:
    free(rt); free(r); free(t); free(p); free(v); free(ph); free(sh); sys_close(&S);
    return out;
}

/* lis_bicg, lis_solver_bicg.c:135-268: the dual recurrences with A^T (real build: conj is the identity);
 * z/ztld double as q/qtld (work[2], work[3], :160-165) */
orc_result orc_bicg(int n, const int *ptr, const int *idx, const double *val,
                    const double *b, double *x, int precon, double tol, int maxiter,
                    int init_zero, double *rhistory)
{
    orc_result out = {0, 0, 0.0};
    orc_sys S; sys_open(&S, n, ptr, idx, val, precon);
    double *r = vec_new(n), *rt = vec_new(n), *z = vec_new(n), *zt = vec_new(n), *p = vec_new(n), *pt = vec_new(n);
    double *q = z, *qt = zt;
    double bnrm, nrm = 0.0, rho_old = 1.0;
    if (rhistory) rhistory[0] = 1.0;
    if (init_zero) memset(x, 0, (size_t)n * sizeof(double));
    if (initial_residual(&S, b, x, init_zero, r, tol, &bnrm, &out)) goto done;
    memcpy(rt, r, (size_t)n * sizeof(double));
    int it;
    for (it = 1; it <= maxiter; it++) {
        sys_psolve(&S, r, z);
        sys_psolve(&S, rt, zt);                      /* M^-H = M^-1 for none / Jacobi (lis_precon_jacobi.c:188-191) */
        const double rho = orc_dot(n, rt, z);
        if (rho == 0.0) { out.retcode = 2; out.iter = it; out.resid = nrm; goto done; }
        const double beta = rho / rho_old;
        orc_xpay(n, z, beta, p);
        sys_matvec(&S, p, q);
        orc_xpay(n, zt, beta, pt);
        orc_spmvh_csr(n, n, S.ptr, S.idx, S.val, pt, qt);
        const double d = orc_dot(n, pt, q);
        if (d == 0.0) { out.retcode = 2; out.iter = it; out.resid = nrm; goto done; }
        const double alpha = rho / d;
        orc_axpy(n, alpha, p, x);
        orc_axpy(n, -alpha, q, r);
        nrm = orc_nrm2(n, r) * bnrm;
        if (rhistory) rhistory[it] = nrm;
        if (tol >= nrm) { out.retcode = 0; out.iter = it; out.resid = nrm; goto done; }
        orc_axpy(n, -alpha, qt, rt);
        rho_old = rho;
    }
    out.retcode = 4; out.iter = it; out.resid = nrm;
done:
    free(r); free(rt); free(z); free(zt); free(p); free(pt); sys_close(&S);
    return out;
}

orc_result orc_gmres(int n, const int *ptr, const int *idx, const double *val,
                     const double *b, double *x, int precon, double tol, int maxiter,
                     int restart, int init_zero, double *rhistory)
{
    orc_result out = {0, 0, 0.0};
    orc_sys S; sys_open(&S, n, ptr, idx, val, precon);
    const int m = restart, ld = m + 1;                /* Hessenberg column stride, then cs/sn rows */
    double *h = (double *)calloc((size_t)(ld + 1) * (ld + 2), sizeof(double));
    double *g = (double *)calloc((size_t)ld + 1, sizeof(double));           /* "s" of the reference */
    double *r = vec_new(n), *z = vec_new(n);
    double **V = (double **)malloc((size_t)(m + 1) * sizeof(double *));
    for (int j = 0; j <= m; j++) V[j] = vec_new(n);
    const int CS = (m + 1) * ld, SN = (m + 2) * ld;
    double bnrm, nrm = 0.0;
    if (rhistory) rhistory[0] = 1.0;
    if (init_zero) memset(x, 0, (size_t)n * sizeof(double));
    /* :187-190 computes M^-1(b - A x) into v0, then the initial-residual helper overwrites v0 with the
     * unpreconditioned residual (:193); only the latter survives. */
    if (initial_residual(&S, b, x, init_zero, V[0], tol, &bnrm, &out)) goto done;
    int it = 0;
    while (it < maxiter) {
        double rn = orc_nrm2(n, V[0]);
        orc_scale(n, 1.0 / rn, V[0]);
        for (int j = 0; j <= m; j++) g[j] = 0.0;
        g[0] = rn;
        int i = 0, ii = 0, i1 = 0;
        do {
            it++; i++;
            ii = i - 1; i1 = i;
            double *hc = h + (size_t)ii * ld;
            sys_psolve(&S, V[ii], z);
            sys_matvec(&S, z, V[i1]);
            for (int k = 0; k < i; k++) {
                double t = orc_dot(n, V[i1], V[k]);
                hc[k] = t;
                orc_axpy(n, -t, V[k], V[i1]);
            }
            double t = orc_nrm2(n, V[i1]);
            hc[i1] = t;
            orc_scale(n, 1.0 / t, V[i1]);
            for (int k = 1; k <= ii; k++) {
                int jj = k - 1;
                double tt = hc[jj];
                double aa = h[jj + CS] * tt;  aa += h[jj + SN] * hc[k];
                double bb = -h[jj + SN] * tt; bb += h[jj + CS] * hc[k];
                hc[jj] = aa; hc[k] = bb;
            }
            double aa = hc[ii], bb = hc[i1];
            double a2 = aa * aa, b2 = bb * bb;
            double rr = sqrt(a2 + b2);
            if (rr == 0.0) rr = 1.0e-17;
            h[ii + CS] = aa / rr;
            h[ii + SN] = bb / rr;
            g[i1] = -h[ii + SN] * g[ii];
            g[ii] =  h[ii + CS] * g[ii];
            aa  = h[ii + CS] * hc[ii];
            aa += h[ii + SN] * hc[i1];
            hc[ii] = aa;
            nrm = fabs(g[i1]) * bnrm;
            if (rhistory) rhistory[it] = nrm;
            if (tol >= nrm) break;
        } while (i < m && it < maxiter);

        /* back substitution on the (ii+1) x (ii+1) upper triangle */
        g[ii] = g[ii] / h[ii + (size_t)ii * ld];
        for (int k = 1; k <= ii; k++) {
            int jj = ii - k;
            double t = g[jj];
            for (int j = jj + 1; j <= ii; j++) t -= h[jj + (size_t)j * ld] * g[j];
            g[jj] = t / h[jj + (size_t)jj * ld];
        }
        for (int k = 0; k < n; k++) z[k] = g[0] * V[0][k];
        for (int j = 1; j <= ii; j++) orc_axpy(n, g[j], V[j], z);
        sys_psolve(&S, z, r);
        orc_axpy(n, 1.0, r, x);
        if (tol >= nrm) { out.retcode = 0; out.iter = it; out.resid = nrm; goto done; }
        for (int j = 1; j <= i; j++) {
            int jj = i1 - j + 1;
            g[jj - 1] = -h[jj - 1 + SN] * g[jj];
            g[jj]     =  h[jj - 1 + CS] * g[jj];
        }
        for (int j = 0; j <= i1; j++) {
            double t = g[j];
            if (j == 0) t = t - 1.0;
            orc_axpy(n, t, V[j], V[0]);
        }
    }
    out.retcode = 4; out.iter = it + 1; out.resid = nrm;
done:
    for (int j = 0; j <= m; j++) free(V[j]);
    free(V); free(r); free(z); free(h); free(g); sys_close(&S);
    return out;
}
