/*
 * lis_oracle.h -- CPU oracle for the Lis SpMV + Krylov hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lis_amd/ may include, link or call this.
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Every function is a plain-C, single-thread, no-FMA restatement of the arithmetic the reference
 * (anishida/lis 2.1.11, /root/reference) performs on this path, in the SAME summation order, so a
 * result computed here is bit-identical to the reference built with its own defaults and run at
 * OMP_NUM_THREADS=1.  Pinned in tests/test_oracle_vs_ref.py against oracle/_ref/liblis_ref.so (the
 * reference compiled from its own sources) and against the committed fixtures in tests/golden/.
 *
 * All arrays are plain host arrays: int32 indices (LIS_INT, include/lis.h:461 of the reference),
 * f64 scalars (LIS_SCALAR, lis.h:446).
 */
#ifndef LIS_ORACLE_H
#define LIS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs ------------------------------------------------------- */
/* 1-D 3-pt Poisson rows [is,ie) of a gn x gn matrix, entry order (i-1, i+1, i): test/spmvtest1.c:139-146.
 * Columns are global.  Returns nnz written. */
int orc_gen_poisson1d(int gn, int is, int ie, int *ptr, int *idx, double *val);
/* 3-D 7-pt Poisson on an l x m x n grid, rows [is,ie); entry order (-mn,+mn,-n,+n,-1,+1,diag) as
 * test/test3.c:114-127; if sort_cols!=0 each row is re-ordered by ascending column as
 * test/spmvtest3.c:192-195 does.  Returns nnz written. */
int orc_gen_poisson3d(int l, int m, int n, int is, int ie, int sort_cols,
                      int *ptr, int *idx, double *val);

/* ---- SpMV, one function per storage format ---------------------------------------------- */
/* src/matvec/lis_matvec_csr.c:90-110 (unsplit branch) */
void orc_spmv_csr(int n, const int *ptr, const int *idx, const double *val,
                  const double *x, double *y);
/* src/matvec/lis_matvec_csc.c:128-144 (serial branch == OpenMP branch at 1 thread) */
void orc_spmv_csc(int n, int np, const int *ptr, const int *idx, const double *val,
                  const double *x, double *y);
/* src/matvec/lis_matvec_ell.c:113-128; column-major val[j*n+i] */
void orc_spmv_ell(int n, int maxnzr, const int *idx, const double *val,
                  const double *x, double *y);
/* src/matvec/lis_matvec_dia.c:148-172 with `nchunks` static row chunks (the reference's layout
 * is blocked by its OpenMP team size: value[is*nnd + d*(ie-is) + i-is]); nchunks=1 -> val[d*n+i] */
void orc_spmv_dia(int n, int nnd, int nchunks, const int *off, const double *val,
                  const double *x, double *y);
/* src/matvec/lis_matvec_jad.c:170-196 with `nchunks` row chunks; ptr has nchunks*(maxnzr+1) entries */
void orc_spmv_jad(int n, int maxnzr, int nchunks, const int *perm, const int *ptr,
                  const int *idx, const double *val, const double *x, double *y);
/* src/matvec/lis_matvec_bsr.c:120-148 (generic) == the unrolled RxC kernels :152-858 row by row.
 * y must have room for nr*bnr entries. */
void orc_spmv_bsr(int n, int nr, int bnr, int bnc, const int *bptr, const int *bidx,
                  const double *val, const double *x, double *y);

/* ---- y = A^T x: lis_matvech_<fmt>, scatter order of the reference at one thread ------------ */
void orc_spmvh_csr(int n, int np, const int *ptr, const int *idx, const double *val,
                   const double *x, double *y);                      /* lis_matvec_csr.c:213-250 */
void orc_spmvh_csc(int np, const int *ptr, const int *idx, const double *val,
                   const double *x, double *y);                      /* lis_matvec_csc.c:176-190 */
void orc_spmvh_ell(int n, int np, int maxnzr, const int *idx, const double *val,
                   const double *x, double *y);                      /* lis_matvec_ell.c:181-215 */
void orc_spmvh_dia(int n, int np, int nnd, const int *off, const double *val,
                   const double *x, double *y);                      /* lis_matvec_dia.c:262-310 */
void orc_spmvh_jad(int n, int np, int maxnzr, const int *perm, const int *ptr, const int *idx,
                   const double *val, const double *x, double *y);   /* lis_matvec_jad.c:545-580 */
void orc_spmvh_bsr(int nr, int bnr, int bnc, const int *bptr, const int *bidx, const double *val,
                   const double *x, double *y, int ylen);            /* lis_matvec_bsr.c:935-957 */

/* ---- CSR -> other formats (layouts the reference defines; 1-thread semantics) -------------- */
int  orc_ell_maxnzr(int n, const int *ptr);
/* src/matrix/lis_matrix_ell.c:1018-1043 */
void orc_csr2ell(int n, const int *ptr, const int *idx, const double *val,
                 int maxnzr, int *eidx, double *eval);
/* src/matrix/lis_matrix_csc.c:1044-1066 (serial branch) */
void orc_csr2csc(int n, int np, const int *ptr, const int *idx, const double *val,
                 int *cptr, int *cidx, double *cval);
/* src/matrix/lis_matrix_dia.c:1217-1283.  NOTE the reference sorts the INPUT rows by column first
 * (lis_matrix_sort_csr, :1217); callers pass column-sorted rows.  Pass off=NULL to only count.
 * Returns nnd. */
int  orc_csr2dia(int n, int nnz, const int *ptr, const int *idx, const double *val,
                 int *off, double *dval);
/* src/matrix/lis_matrix_jad.c:1650-1735 at one chunk, including the reference's own
 * (unstable) descending quicksort src/system/lis_sort.c:249-276 that fixes the tie order of perm. */
int  orc_jad_maxnzr(int n, const int *ptr);
void orc_csr2jad(int n, const int *ptr, const int *idx, const double *val, int maxnzr,
                 int *perm, int *jptr, int *jidx, double *jval);
/* src/matrix/lis_matrix_bsr.c:351-552 (non-MPI).  Pass bidx=NULL to only fill bptr and count.
 * Returns bnnz. */
int  orc_csr2bsr(int n, const int *ptr, const int *idx, const double *val, int bnr, int bnc,
                 int *bptr, int *bidx, double *bval);
/* src/matrix/lis_matrix_csr.c:547-558 */
void orc_csr_diagonal(int n, const int *ptr, const int *idx, const double *val, double *d);

/* ---- vector kernels ----------------------------------------------------------------------- */
double orc_dot (int n, const double *x, const double *y);           /* lis_vector_ops.c:97-107 @1thr */
double orc_nrm2(int n, const double *x);                            /* lis_vector_ops.c:241-266      */
double orc_nrm1(int n, const double *x);                            /* lis_vector_ops.c:278-342      */
void orc_axpy (int n, double a, const double *x, double *y);        /* lis_vector_opv.c:174-177 y+=a*x */
void orc_xpay (int n, const double *x, double a, double *y);        /* :214-217  y = x + a*y          */
void orc_axpyz(int n, double a, const double *x, const double *y, double *z); /* :253-256 z=a*x+y      */
void orc_scale(int n, double a, double *x);                         /* :285-288                       */
void orc_pmul (int n, const double *x, const double *y, double *z); /* :325-328                       */
void orc_reciprocal(int n, double *x);                              /* :458-461                       */

/* ---- Krylov loops on a CSR matrix ----------------------------------------------------------- */
typedef struct {
    int    iter;        /* solver->iter   */
    int    retcode;     /* 0 success, 2 breakdown, 4 maxiter  (lis.h:1052-1063) */
    double resid;       /* solver->resid (relative, last computed) */
} orc_result;

#define ORC_PRECON_NONE   0
#define ORC_PRECON_JACOBI 1

/* x is the initial guess on entry when init_zero==0 (lis_solver.c:561-592) and the solution on exit.
 * rhistory (maxiter+2 entries, may be NULL) receives rhistory[iter] as with `-print mem`. */
/* src/solver/lis_solver_cg.c:129-235 */
orc_result orc_cg(int n, const int *ptr, const int *idx, const double *val,
                  const double *b, double *x, int precon, double tol, int maxiter,
                  int init_zero, double *rhistory);
/* src/solver/lis_solver_bicgstab.c:137-315 */
orc_result orc_bicgstab(int n, const int *ptr, const int *idx, const double *val,
                        const double *b, double *x, int precon, double tol, int maxiter,
                        int init_zero, double *rhistory);
/* src/solver/lis_solver_gmres.c:135-342 */
orc_result orc_bicg(int n, const int *ptr, const int *idx, const double *val,
                    const double *b, double *x, int precon, double tol, int maxiter,
                    int init_zero, double *rhistory);               /* lis_solver_bicg.c:135-268 */
orc_result orc_gmres(int n, const int *ptr, const int *idx, const double *val,
                     const double *b, double *x, int precon, double tol, int maxiter,
                     int restart, int init_zero, double *rhistory);

#ifdef __cplusplus
}
#endif
#endif
