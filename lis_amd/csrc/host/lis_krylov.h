/*
 * lis_krylov.h -- what every Krylov loop of liblis_amd shares: the device-side solve context (work vectors in
 * HBM, preconditioner diagonal, tolerances), the residual bookkeeping of lis_solver.c:957-1091 / :1792-1812,
 * and the error-unwinding macros.  Included by lis_solver.c (CG, BiCG, BiCGSTAB, GMRES with fused passes) and
 * lis_solver_more.c (the other short-recurrence solvers, one kernel per reference call).
 */
#ifndef LIS_AMD_KRYLOV_H
#define LIS_AMD_KRYLOV_H
#include "lis_internal.h"

/* ------------------------------------------------------------------ the device-side solve context */
typedef struct {
	LIS_SOLVER s;
	LIS_MATRIX A;
	int n;
	size_t len;              /* doubles per work vector: np + pad + slack (ghost slots for the halo) */
	double *b, *x;           /* HBM */
	double *dinv;            /* Jacobi 1/diag in HBM, NULL for none */
	int duniform; double dconst;   /* every dinv[i] is the double dconst (a constant diagonal): the fused CG passes take the scalar */
	double **work; int nwork;
	double bnrm, tol;
	int output, maxiter;
} ctx_t;

static inline LIS_INT work_alloc(ctx_t *c, int count)
{
	c->work = (double **)calloc((size_t)count, sizeof(double *));
	c->nwork = count;
	for (int i = 0; i < count; i++) {
		LISCHK(lisd_pool_get(c->len * sizeof(double), (void **)&c->work[i]));
		HIPCHK(liship_memset(c->work[i], 0, c->len * sizeof(double), lisg.stream));
	}
	return LIS_SUCCESS;
}
static inline void work_free(ctx_t *c)
{
	for (int i = 0; i < c->nwork; i++) lisd_pool_put(c->work[i], c->len * sizeof(double));
	free(c->work); c->work = NULL; c->nwork = 0;
}

#define K(call) HIPCHK(call)
static inline LIS_INT d_copy(ctx_t *c, const double *src, double *dst) { K(liship_memcpy_d2d(dst, src, sizeof(double) * (size_t)c->n, lisg.stream)); return LIS_SUCCESS; }
static inline LIS_INT d_psolve(ctx_t *c, const double *r, double *z)
{	/* none: copy (lis_precon.c:365-384); Jacobi: z = r .* dinv (lis_precon_jacobi.c:121-124) */
	if (c->dinv) { K(liship_pmul_f64(c->n, r, c->dinv, z, lisg.stream)); return LIS_SUCCESS; }
	return d_copy(c, r, z);
}
static inline LIS_INT d_matvec(ctx_t *c, double *x, double *y) { return lisd_spmv(c->A, x, y); }
static inline LIS_INT d_resid(ctx_t *c, const double *r, double *nrm)
{	/* lis_solver_get_residual_nrm2_r (lis_solver.c:1792) / _nrm1_b (:1804) */
	if (c->s->options[LIS_OPTIONS_CONV_COND] == LIS_CONV_COND_NRM1_B) return lisd_nrm1(c->n, r, nrm);
	LISCHK(lisd_nrm2(c->n, r, nrm));
	*nrm = *nrm * c->bnrm;
	return LIS_SUCCESS;
}
static inline void note(ctx_t *c, LIS_INT iter, double nrm)
{
	if (!c->output) return;
	if ((c->output & LIS_PRINT_MEM) && iter <= c->maxiter + 1) c->s->rhistory[iter] = nrm;   /* maxiter + 2 slots (IDR(1) can step past) */
	if (c->output & LIS_PRINT_OUT) lis_printf(LIS_COMM_WORLD, "iteration: %5d  relative residual = %e\n", (int)iter, nrm);
}

/* r = b - A x (or b when x0 = 0), scaling 1/||r||, early exit when already converged: lis_solver.c:957-1091.
 * returns 1 when the caller must stop (converged), 0 to iterate, <0 on error (-err) */
static inline int initial_residual(ctx_t *c, double *r)
{
	LIS_SOLVER s = c->s;
	const int conv = s->options[LIS_OPTIONS_CONV_COND];
	const double tol = s->params[LIS_PARAMS_RESID - LIS_OPTIONS_LEN], tol_w = s->params[LIS_PARAMS_RESID_WEIGHT - LIS_OPTIONS_LEN];
	LIS_INT err = 0;
	if (!s->options[LIS_OPTIONS_INITGUESS_ZEROS]) {
		err = d_matvec(c, c->x, r);
		if (!err && liship_xpay_f64(c->n, c->b, -1.0, r, lisg.stream)) err = LIS_ERR_NOT_IMPLEMENTED;
	} else err = d_copy(c, c->b, r);
	if (err) return -(int)err;
	double nrm = 0.0, bn = 0.0;
	switch (conv) {
	case LIS_CONV_COND_NRM2_R: err = lisd_nrm2(c->n, r, &nrm); bn = nrm; s->tol = tol; break;
	case LIS_CONV_COND_NRM2_B: err = lisd_nrm2(c->n, r, &nrm); if (!err) err = lisd_nrm2(c->n, c->b, &bn); s->tol = tol; break;
	default:                   err = lisd_nrm1(c->n, r, &nrm); if (!err) err = lisd_nrm1(c->n, c->b, &bn); s->tol = bn * tol_w + tol; break;
	}
	if (err) return -(int)err;
	s->tol_switch = s->params[LIS_PARAMS_SWITCH_RESID - LIS_OPTIONS_LEN];
	bn = (bn == 0.0) ? 1.0 : 1.0 / bn;
	s->bnrm = bn; c->bnrm = bn; c->tol = s->tol;
	nrm = nrm * bn;
	if (nrm <= fabs(tol)) { s->retcode = LIS_SUCCESS; s->iter = 1; s->resid = nrm; return 1; }
	return 0;
}

#define TRY(expr) do { LIS_INT e__ = (expr); if (e__) { err = e__; goto done; } } while (0)
#define KTRY(call) do { int rc__ = (call); if (rc__) { err = lisi_hip_error(__FILE__, __func__, __LINE__, rc__); goto done; } } while (0)

/* lis_solver_more.c */
LIS_INT lisk_cgs(ctx_t *c);
LIS_INT lisk_cr(ctx_t *c);
LIS_INT lisk_gpbicg(ctx_t *c);
LIS_INT lisk_tfqmr(ctx_t *c);
LIS_INT lisk_bicgsafe(ctx_t *c);
LIS_INT lisk_orthomin(ctx_t *c);
LIS_INT lisk_gpbicr(ctx_t *c);
LIS_INT lisk_bicr(ctx_t *c);
LIS_INT lisk_crs(ctx_t *c);
LIS_INT lisk_bicrstab(ctx_t *c);
LIS_INT lisk_bicrsafe(ctx_t *c);
LIS_INT lisk_fgmres(ctx_t *c);
LIS_INT lisk_minres(ctx_t *c);
LIS_INT lisk_idrs(ctx_t *c);
LIS_INT lisk_bicgstabl(ctx_t *c);
LIS_INT lisk_jacobi(ctx_t *c);

#endif
