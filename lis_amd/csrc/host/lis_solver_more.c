/*
 * lis_solver_more.c -- the other short-recurrence Krylov solvers of Lis on the same HBM work vectors and the
 * same kernels: CGS, CR, GPBiCG, TFQMR, BiCGSafe, Orthomin(m).
 *
 * Each loop issues the reference's vector operations in the reference's order, one kernel per call
 * (element-wise results are bit-identical, reductions are the deterministic trees of vector_ops.hip), so the
 * recurrences see the same numbers as the CPU path up to the reduction order.  No pass fusion here: these are
 * coverage, CG / BiCG / BiCGSTAB / GMRES in lis_solver.c are the tuned ones.
 *   CGS       src/solver/lis_solver_cgs.c:134-276        CR         lis_solver_cg.c:821-940
 *   GPBiCG    lis_solver_gpbicg.c:145-351                TFQMR      lis_solver_qmr.c:113-299
 *   BiCGSafe  lis_solver_bicgsafe.c:145-322              Orthomin   lis_solver_orthomin.c:124-250
 * and the conjugate-residual family + the rest that needs no new kernel:
 *   BiCR      lis_solver_bicg.c:788-926 (uses A^T)       CRS        lis_solver_cgs.c:805-938
 *   BiCRSTAB  lis_solver_bicgstab.c:951-1098             GPBiCR     lis_solver_gpbicg.c:1349-1556
 *   BiCRSafe  lis_solver_bicgsafe.c:1048-1227            FGMRES(m)  lis_solver_gmres.c:1128-1303
 *   MINRES    lis_solver_minres.c:121-258                COCG/COCR  lis_solver_cg.c:632-739, :1155-1274 (real build:
 *                                                                   the arithmetic of CG / CR)
 */
#include "lis_krylov.h"

#define AXPY(a, x, y)      KTRY(liship_axpy_f64(n, (a), (x), (y), lisg.stream))          /* y += a x     */
#define XPAY(x, a, y)      KTRY(liship_xpay_f64(n, (x), (a), (y), lisg.stream))          /* y = x + a y  */
#define AXPYZ(a, x, y, z)  KTRY(liship_axpyz_f64(n, (a), (x), (y), (z), lisg.stream))    /* z = a x + y  */
#define SCALE(a, x)        KTRY(liship_scale_f64(n, (a), (x), lisg.stream))
#define COPY(src, dst)     TRY(d_copy(c, (src), (dst)))
#define DOT(x, y, out)     TRY(lisd_dot(n, (x), (y), (out)))
#define MATVEC(x, y)       TRY(d_matvec(c, (x), (y)))
#define PSOLVE(r, z)       TRY(d_psolve(c, (r), (z)))
#define RESID(r, out)      TRY(d_resid(c, (r), (out)))
#define FINISH(code) do { s->retcode = (code); s->iter = iter; s->resid = nrm2; err = ((code) == LIS_SUCCESS) ? 0 : (code); goto done; } while (0)

LIS_INT lisk_cgs(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old = 1.0, alpha, beta, d1;
	TRY(work_alloc(c, 7));
	double *rtld = c->work[0], *r = c->work[1], *p = c->work[2], *phat = c->work[3], *q = c->work[4],
	       *qhat = c->work[5], *u = c->work[5], *uhat = c->work[6], *vhat = c->work[6];      /* aliases as in the reference */
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, rtld);
	for (iter = 1; iter <= c->maxiter; iter++) {
		DOT(rtld, r, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = rho / rho_old;
		AXPYZ(beta, q, r, u);                 /* u = r + beta q */
		XPAY(q, beta, p);                     /* p = u + beta (q + beta p) */
		XPAY(u, beta, p);
		PSOLVE(p, phat);
		MATVEC(phat, vhat);
		DOT(rtld, vhat, &d1);
		if (d1 == 0.0) FINISH(LIS_BREAKDOWN);
		alpha = rho / d1;
		AXPYZ(-alpha, vhat, u, q);            /* q = u - alpha vhat */
		AXPYZ(1.0, u, q, phat);               /* phat = u + q */
		PSOLVE(phat, uhat);
		AXPY(alpha, uhat, c->x);
		MATVEC(uhat, qhat);
		AXPY(-alpha, qhat, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_cr(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, alpha, beta, dot_rq, dot_zq;
	TRY(work_alloc(c, 6));
	double *z = c->work[0], *q = c->work[1], *r = c->work[2], *p = c->work[3], *qtld = c->work[4], *az = c->work[5];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	PSOLVE(r, p);
	MATVEC(p, q);
	COPY(p, z);
	for (iter = 1; iter <= c->maxiter; iter++) {
		PSOLVE(q, qtld);
		DOT(qtld, q, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		DOT(r, qtld, &dot_rq);
		alpha = dot_rq / rho;
		AXPY(alpha, p, c->x);
		AXPY(-alpha, q, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		AXPY(-alpha, qtld, z);
		MATVEC(z, az);
		DOT(az, qtld, &dot_zq);
		beta = -dot_zq / rho;
		XPAY(z, beta, p);
		XPAY(az, beta, q);
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

/* the 2x2 normal equations both GPBiCG and BiCGSafe solve for (qsi, eta) */
static void qsi_eta(int first, const double *t, double *qsi, double *eta)
{
	if (first) { *qsi = t[1] / t[4]; *eta = 0.0; return; }
	const double tmp = t[4] * t[0] - t[3] * t[3];
	*qsi = (t[0] * t[1] - t[2] * t[3]) / tmp;
	*eta = (t[4] * t[2] - t[3] * t[1]) / tmp;
}

static LIS_INT gpbi(ctx_t *c, int cr);
LIS_INT lisk_gpbicg(ctx_t *c) { return gpbi(c, 0); }
LIS_INT lisk_gpbicr(ctx_t *c) { return gpbi(c, 1); }

/* GPBiCG, and GPBiCR (cr): the same recurrences with the shadow residual A^T r0 and the dots taken against the
 * preconditioned vectors (lis_solver_gpbicg.c:1349-1556) */
static LIS_INT gpbi(ctx_t *c, int cr)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old, alpha, beta = 0.0, qsi, eta, t5[5];
	TRY(work_alloc(c, 14));
	double *rtld = c->work[0], *r = c->work[1], *mr = c->work[2], *p = c->work[3], *ap = c->work[4], *map = c->work[5],
	       *t = c->work[6], *mt = c->work[7], *amt = c->work[8], *u = c->work[9], *y = c->work[10], *w = c->work[11],
	       *z = c->work[12], *mt_old = c->work[13];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	if (cr) { COPY(r, p); TRY(lisd_spmv_t(c->A, p, rtld)); } else COPY(r, rtld);
	PSOLVE(r, p);
	DOT(rtld, cr ? p : r, &rho_old);
	for (iter = 1; iter <= c->maxiter; iter++) {
		MATVEC(p, ap);
		PSOLVE(ap, map);
		DOT(rtld, cr ? map : ap, &t5[0]);
		if (t5[0] == 0.0) FINISH(LIS_BREAKDOWN);
		alpha = rho_old / t5[0];
		AXPYZ(-1.0, w, ap, y);                /* y = t - r + alpha (ap - w) */
		XPAY(t, alpha, y);
		AXPY(-1.0, r, y);
		AXPYZ(-alpha, ap, r, t);              /* t = r - alpha ap */
		RESID(t, &nrm2);
		if (nrm2 <= c->tol) {
			note(c, iter, nrm2);
			AXPY(alpha, p, c->x);
			FINISH(LIS_SUCCESS);
		}
		AXPYZ(-alpha, map, mr, mt);
		MATVEC(mt, amt);
		DOT(y, y, &t5[0]); DOT(amt, t, &t5[1]); DOT(y, t, &t5[2]); DOT(amt, y, &t5[3]); DOT(amt, amt, &t5[4]);
		qsi_eta(iter == 1, t5, &qsi, &eta);
		XPAY(mt_old, beta, u);                /* u = qsi map + eta (mt_old - mr + beta u) */
		AXPY(-1.0, mr, u);
		SCALE(eta, u);
		AXPY(qsi, map, u);
		SCALE(eta, z);                        /* z = qsi mr + eta z - alpha u */
		AXPY(qsi, mr, z);
		AXPY(-alpha, u, z);
		AXPY(alpha, p, c->x);
		AXPY(1.0, z, c->x);
		AXPYZ(-qsi, amt, t, r);               /* r = t - eta y - qsi amt */
		AXPY(-eta, y, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		PSOLVE(r, mr);
		DOT(rtld, cr ? mr : r, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = (rho / rho_old) * (alpha / qsi);
		AXPYZ(beta, ap, amt, w);              /* w = amt + beta ap */
		AXPY(-1.0, u, p);                     /* p = mr + beta (p - u) */
		XPAY(mr, beta, p);
		COPY(mt, mt_old);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_tfqmr(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 1;
	const int n = c->n;
	double nrm2 = 0.0, tau, rho, rhoold, theta = 0.0, eta = 0.0, beta, alpha, w, ww, wold, sd, cc;
	TRY(work_alloc(c, 9));
	double *r = c->work[0], *rtld = c->work[1], *u = c->work[2], *p = c->work[3], *d = c->work[4], *t = c->work[5],
	       *t1 = c->work[6], *q = c->work[7], *v = c->work[8];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, rtld);
	COPY(r, p);
	COPY(r, u);
	PSOLVE(p, t);
	MATVEC(t, v);
	DOT(r, rtld, &rhoold);
	TRY(lisd_nrm2(n, r, &tau));
	wold = tau;
	while (iter <= c->maxiter) {
		DOT(v, rtld, &sd);
		if (sd == 0.0) FINISH(LIS_BREAKDOWN);
		alpha = rhoold / sd;
		AXPYZ(-alpha, v, u, q);
		AXPYZ(1.0, u, q, t);
		PSOLVE(t, t1);
		MATVEC(t1, v);
		AXPY(-alpha, v, r);
		TRY(lisd_nrm2(n, r, &w));
		for (int m = 0; m < 2; m++) {
			if (m == 0) { ww = sqrt(w * wold); XPAY(u, theta * theta * eta / alpha, d); }
			else        { ww = w;              XPAY(q, theta * theta * eta / alpha, d); }
			theta = ww / tau;
			cc = 1.0 / sqrt(1.0 + theta * theta);
			eta = cc * cc * alpha;
			tau = tau * theta * cc;
			PSOLVE(d, t1);
			AXPY(eta, t1, c->x);
			nrm2 = tau * sqrt(1.0 + m) * c->bnrm;
			if (m == 0) note(c, iter, nrm2);
			if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		}
		DOT(r, rtld, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = rho / rhoold;
		AXPYZ(beta, q, r, u);
		XPAY(q, beta, p);
		XPAY(u, beta, p);
		PSOLVE(p, t1);
		MATVEC(t1, v);
		rhoold = rho;
		wold = w;
		iter++;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_bicgsafe(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old, alpha, beta = 0.0, qsi, eta, t5[5];
	TRY(work_alloc(c, 12));
	double *rtld = c->work[0], *r = c->work[1], *mr = c->work[2], *amr = c->work[3], *p = c->work[4], *ap = c->work[5],
	       *t = c->work[6], *mt = c->work[7], *y = c->work[8], *u = c->work[9], *z = c->work[10], *au = c->work[11];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, rtld);
	PSOLVE(r, mr);
	MATVEC(mr, amr);
	DOT(rtld, r, &rho_old);
	COPY(amr, ap);
	COPY(mr, p);
	for (iter = 1; iter <= c->maxiter; iter++) {
		DOT(rtld, ap, &t5[0]);
		alpha = rho_old / t5[0];
		DOT(y, y, &t5[0]); DOT(amr, r, &t5[1]); DOT(y, r, &t5[2]); DOT(amr, y, &t5[3]); DOT(amr, amr, &t5[4]);
		qsi_eta(iter == 1, t5, &qsi, &eta);
		COPY(y, t);                           /* t = qsi ap + eta y */
		SCALE(eta, t);
		AXPY(qsi, ap, t);
		PSOLVE(t, mt);
		XPAY(mt, eta * beta, u);              /* u = mt + eta beta u */
		MATVEC(u, au);
		SCALE(eta, z);                        /* z = qsi mr + eta z - alpha u */
		AXPY(qsi, mr, z);
		AXPY(-alpha, u, z);
		SCALE(eta, y);                        /* y = qsi amr + eta y - alpha au */
		AXPY(qsi, amr, y);
		AXPY(-alpha, au, y);
		AXPY(alpha, p, c->x);
		AXPY(1.0, z, c->x);
		AXPY(-alpha, ap, r);
		AXPY(-1.0, y, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		DOT(rtld, r, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = (rho / rho_old) * (alpha / qsi);
		PSOLVE(r, mr);
		MATVEC(mr, amr);
		AXPY(-1.0, u, p);                     /* p = mr + beta (p - u) */
		XPAY(mr, beta, p);
		AXPY(-1.0, au, ap);                   /* ap = amr + beta (ap - au) */
		XPAY(amr, beta, ap);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_orthomin(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 1;
	const int n = c->n, m = s->options[LIS_OPTIONS_RESTART];
	double nrm2 = 0.0, alpha, beta;
	double *dotsave = (double *)calloc((size_t)m + 1, sizeof(double));
	if (!dotsave) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", m + 1); goto done; }
	TRY(work_alloc(c, 2 + 3 * (m + 1)));
	double *r = c->work[0], *rtld = c->work[1], **p = &c->work[2], **ap = &c->work[(m + 1) + 2], **aptld = &c->work[2 * (m + 1) + 2];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	PSOLVE(r, rtld);                              /* the M != NULL form of the initial residual (lis_solver.c:1083-1087) */
	while (iter <= c->maxiter) {
		const int ip = (iter - 1) % (m + 1);
		COPY(rtld, p[ip]);
		MATVEC(p[ip], ap[ip]);
		PSOLVE(ap[ip], aptld[ip]);
		const int lmax = m < iter - 1 ? m : iter - 1;
		for (int l = 1; l <= lmax; l++) {
			const int ip0 = (ip + m + 1 - l) % (m + 1);
			DOT(aptld[ip], aptld[ip0], &beta);
			beta = -beta * dotsave[l - 1];
			AXPY(beta, p[ip0], p[ip]);
			AXPY(beta, ap[ip0], ap[ip]);
			AXPY(beta, aptld[ip0], aptld[ip]);
		}
		for (int l = m - 1; l > 0; l--) dotsave[l] = dotsave[l - 1];
		DOT(aptld[ip], aptld[ip], &dotsave[0]);
		if (dotsave[0] == 0.0) FINISH(LIS_BREAKDOWN);
		dotsave[0] = 1.0 / dotsave[0];
		DOT(rtld, aptld[ip], &alpha);
		alpha = alpha * dotsave[0];
		AXPY(alpha, p[ip], c->x);
		AXPY(-alpha, ap[ip], r);
		AXPY(-alpha, aptld[ip], rtld);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		iter++;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	free(dotsave);
	return err;
}

LIS_INT lisk_bicr(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old, alpha, beta, d1;
	TRY(work_alloc(c, 10));
	double *r = c->work[0], *rtld = c->work[1], *z = c->work[2], *ztld = c->work[3], *p = c->work[4], *ptld = c->work[5],
	       *ap = c->work[6], *az = c->work[7], *map = c->work[8], *aptld = c->work[9];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, rtld);
	PSOLVE(r, z);
	PSOLVE(rtld, ztld);                           /* M^-H = M^-1 for none / Jacobi */
	COPY(z, p);
	COPY(ztld, ptld);
	MATVEC(z, ap);
	DOT(ztld, ap, &rho_old);
	for (iter = 1; iter <= c->maxiter; iter++) {
		TRY(lisd_spmv_t(c->A, ptld, aptld));
		PSOLVE(ap, map);
		DOT(aptld, map, &d1);
		if (d1 == 0.0) FINISH(LIS_BREAKDOWN);
		alpha = rho_old / d1;
		AXPY(alpha, p, c->x);
		AXPY(-alpha, ap, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		AXPY(-alpha, aptld, rtld);
		AXPY(-alpha, map, z);
		PSOLVE(rtld, ztld);
		MATVEC(z, az);
		DOT(ztld, az, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = rho / rho_old;
		XPAY(z, beta, p);
		XPAY(ztld, beta, ptld);
		XPAY(az, beta, ap);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_crs(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old = 1.0, alpha, beta, d1;
	TRY(work_alloc(c, 6));
	double *r = c->work[0], *rtld = c->work[1], *p = c->work[2], *z = c->work[3], *u = c->work[3], *uq = c->work[3],
	       *q = c->work[4], *ap = c->work[4], *map = c->work[5], *auq = c->work[5];           /* aliases as in the reference */
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, p);
	TRY(lisd_spmv_t(c->A, p, rtld));              /* shadow residual A^T r0 */
	KTRY(liship_set_all_f64(n, 0.0, q, lisg.stream));
	KTRY(liship_set_all_f64(n, 0.0, p, lisg.stream));
	for (iter = 1; iter <= c->maxiter; iter++) {
		PSOLVE(r, z);
		DOT(rtld, z, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = rho / rho_old;
		AXPYZ(beta, q, z, u);
		XPAY(q, beta, p);
		XPAY(u, beta, p);
		MATVEC(p, ap);
		PSOLVE(ap, map);
		DOT(rtld, map, &d1);
		if (d1 == 0.0) FINISH(LIS_BREAKDOWN);
		alpha = rho / d1;
		AXPYZ(-alpha, map, u, q);
		AXPYZ(1.0, u, q, uq);
		MATVEC(uq, auq);
		AXPY(alpha, uq, c->x);
		AXPY(-alpha, auq, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_bicrstab(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old, alpha, beta, omega, d1, d2;
	TRY(work_alloc(c, 9));
	double *rtld = c->work[0], *r = c->work[1], *sv = c->work[2], *ms = c->work[3], *ams = c->work[4], *p = c->work[5],
	       *ap = c->work[6], *map = c->work[7], *z = c->work[8];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, p);
	TRY(lisd_spmv_t(c->A, p, rtld));
	PSOLVE(r, z);
	COPY(z, p);
	DOT(rtld, z, &rho_old);
	for (iter = 1; iter <= c->maxiter; iter++) {
		MATVEC(p, ap);
		PSOLVE(ap, map);
		DOT(rtld, map, &d1);
		alpha = rho_old / d1;
		AXPYZ(-alpha, ap, r, sv);
		RESID(sv, &nrm2);
		if (nrm2 <= c->tol) {
			note(c, iter, nrm2);
			AXPY(alpha, p, c->x);
			FINISH(LIS_SUCCESS);
		}
		AXPYZ(-alpha, map, z, ms);
		MATVEC(ms, ams);
		DOT(ams, sv, &d1);
		DOT(ams, ams, &d2);
		omega = d1 / d2;
		AXPY(alpha, p, c->x);
		AXPY(omega, ms, c->x);
		AXPYZ(-omega, ams, sv, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		PSOLVE(r, z);
		DOT(rtld, z, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = (rho / rho_old) * (alpha / omega);
		AXPY(-omega, map, p);
		XPAY(z, beta, p);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

LIS_INT lisk_bicrsafe(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	double nrm2 = 0.0, rho, rho_old, alpha, beta = 0.0, qsi, eta, t5[5];
	TRY(work_alloc(c, 13));
	double *rtld = c->work[0], *r = c->work[1], *mr = c->work[2], *amr = c->work[3], *p = c->work[4], *ap = c->work[5],
	       *map = c->work[6], *my = c->work[7], *y = c->work[8], *u = c->work[9], *z = c->work[10], *au = c->work[11],
	       *artld = c->work[12];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r, rtld);
	TRY(lisd_spmv_t(c->A, rtld, artld));
	PSOLVE(r, mr);
	MATVEC(mr, amr);
	DOT(rtld, amr, &rho_old);
	COPY(amr, ap);
	COPY(mr, p);
	for (iter = 1; iter <= c->maxiter; iter++) {
		PSOLVE(ap, map);
		DOT(artld, map, &t5[0]);
		alpha = rho_old / t5[0];
		DOT(y, y, &t5[0]); DOT(amr, r, &t5[1]); DOT(y, r, &t5[2]); DOT(amr, y, &t5[3]); DOT(amr, amr, &t5[4]);
		qsi_eta(iter == 1, t5, &qsi, &eta);
		SCALE(eta * beta, u);                 /* u = qsi map + eta (my + beta u) */
		AXPY(qsi, map, u);
		AXPY(eta, my, u);
		MATVEC(u, au);
		SCALE(eta, z);
		AXPY(qsi, mr, z);
		AXPY(-alpha, u, z);
		SCALE(eta, y);
		AXPY(qsi, amr, y);
		AXPY(-alpha, au, y);
		PSOLVE(y, my);
		AXPY(alpha, p, c->x);
		AXPY(1.0, z, c->x);
		AXPY(-alpha, ap, r);
		AXPY(-1.0, y, r);
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		AXPY(-alpha, map, mr);
		AXPY(-1.0, my, mr);
		MATVEC(mr, amr);
		DOT(rtld, amr, &rho);
		if (rho == 0.0) FINISH(LIS_BREAKDOWN);
		beta = (rho / rho_old) * (alpha / qsi);
		AXPY(-1.0, u, p);
		XPAY(mr, beta, p);
		AXPY(-1.0, au, ap);
		XPAY(amr, beta, ap);
		rho_old = rho;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

/* flexible GMRES: the preconditioned basis z[] is kept, so the update is x += sum y_j z_j.  Its convergence test
 * compares the ABSOLUTE residual |s[i+1]| with the tolerance (lis_solver_gmres.c:1128-1303, no bnrm factor): kept. */
LIS_INT lisk_fgmres(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n, m = s->options[LIS_OPTIONS_RESTART], ld = m + 1;
	const int CS = (m + 1) * ld, SN = (m + 2) * ld;
	double *h = (double *)calloc((size_t)(ld + 1) * (size_t)(ld + 2), sizeof(double));
	double *g = (double *)calloc((size_t)ld + 2, sizeof(double));
	double nrm2 = 0.0, rnorm, bnrm2, t;
	int ii = 0;
	if (!h || !g) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", ld); goto done; }
	TRY(work_alloc(c, 2 * m + 3));
	double **z = &c->work[0], **v = &c->work[m + 1];
	int st = initial_residual(c, v[0]);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	bnrm2 = c->bnrm;
	rnorm = 1.0 / bnrm2;
	while (iter < c->maxiter) {
		SCALE(bnrm2, v[0]);
		for (int j = 0; j <= m + 1; j++) g[j] = 0.0;
		g[0] = rnorm;
		int i = 0;
		do {
			iter++; i++;
			ii = i - 1;
			const int i1 = i;
			double *hc = h + (size_t)ii * ld;
			PSOLVE(v[ii], z[ii]);
			MATVEC(z[ii], v[i1]);
			for (int k = 0; k < i; k++) {
				DOT(v[i1], v[k], &t);
				hc[k] = t;
				AXPY(-t, v[k], v[i1]);
			}
			TRY(lisd_nrm2(n, v[i1], &t));
			hc[i1] = t;
			SCALE(1.0 / t, v[i1]);
			for (int k = 1; k <= ii; k++) {
				const int jj = k - 1;
				const double tt = hc[jj];
				double aa = h[jj + CS] * tt;  aa += h[jj + SN] * hc[k];
				double bb = -h[jj + SN] * tt; bb += h[jj + CS] * hc[k];
				hc[jj] = aa; hc[k] = bb;
			}
			double aa = hc[ii], bb = hc[i1];
			double rr = sqrt(aa * aa + bb * bb);
			if (rr == 0.0) rr = 1.0e-17;
			h[ii + CS] = aa / rr;
			h[ii + SN] = bb / rr;
			g[i1] = -h[ii + SN] * g[ii];
			g[ii] =  h[ii + CS] * g[ii];
			aa  = h[ii + CS] * hc[ii];
			aa += h[ii + SN] * hc[i1];
			hc[ii] = aa;
			nrm2 = fabs(g[i1]);
			note(c, iter, nrm2);
			if (c->tol >= nrm2) break;
		} while (i < m && iter < c->maxiter);
		g[ii] = g[ii] / h[ii + (size_t)ii * ld];
		for (int k = 1; k <= ii; k++) {
			const int jj = ii - k;
			double tt = g[jj];
			for (int j = jj + 1; j <= ii; j++) tt -= h[jj + (size_t)j * ld] * g[j];
			g[jj] = tt / h[jj + (size_t)jj * ld];
		}
		for (int j = 0; j <= ii; j++) AXPY(g[j], z[j], c->x);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		MATVEC(c->x, v[0]);
		XPAY(c->b, -1.0, v[0]);
		TRY(lisd_nrm2(n, v[0], &rnorm));
		bnrm2 = 1.0 / rnorm;
	}
	s->retcode = LIS_MAXITER; s->iter = iter + 1; s->resid = nrm2; err = LIS_MAXITER;
done:
	work_free(c);
	free(h); free(g);
	return err;
}

/* MINRES keeps its own residual bookkeeping: r0 = M^-1 (b - A x) always formed with a product, relative norm
 * r_euc / r0_euc against the raw -tol parameter (lis_solver_minres.c:121-258) */
LIS_INT lisk_minres(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	const double tol = s->params[LIS_PARAMS_RESID - LIS_OPTIONS_LEN];
	double nrm2, alpha, beta2, beta3, gamma1 = 1.0, gamma2 = 1.0, gamma3, delta, eta, sigma1 = 0.0, sigma2 = 0.0, sigma3,
	       rho1, rho2, rho3, r0_euc, r_euc;
	TRY(work_alloc(c, 7));
	double *v1 = c->work[0], *v2 = c->work[1], *v3 = c->work[2], *v4 = c->work[3], *w0 = c->work[4], *w1 = c->work[5], *w2 = c->work[6];
	MATVEC(c->x, v2);
	XPAY(c->b, -1.0, v2);
	PSOLVE(v2, v3);
	COPY(v3, v2);
	TRY(lisd_nrm2(n, v2, &r_euc));
	eta = beta2 = r0_euc = r_euc;
	nrm2 = r_euc / r0_euc;
	for (iter = 1; iter <= c->maxiter; iter++) {
		SCALE(1.0 / beta2, v2);
		MATVEC(v2, v3);
		PSOLVE(v3, v4);
		DOT(v2, v4, &alpha);
		AXPY(-alpha, v2, v4);
		AXPY(-beta2, v1, v4);
		TRY(lisd_nrm2(n, v4, &beta3));
		delta = gamma2 * alpha - gamma1 * sigma2 * beta2;
		rho1 = sqrt(delta * delta + beta3 * beta3);
		rho2 = sigma2 * alpha + gamma1 * gamma2 * beta2;
		rho3 = sigma1 * beta2;
		gamma3 = delta / rho1;
		sigma3 = beta3 / rho1;
		AXPYZ(-rho3, w0, v2, w2);
		AXPY(-rho2, w1, w2);
		SCALE(1.0 / rho1, w2);
		AXPY(gamma3 * eta, w2, c->x);
		r_euc *= fabs(sigma3);
		nrm2 = r_euc / r0_euc;
		note(c, iter, nrm2);
		if (nrm2 <= tol) FINISH(LIS_SUCCESS);
		eta *= -sigma3;
		COPY(v2, v1);
		COPY(v4, v2);
		COPY(w1, w0);
		COPY(w2, w1);
		beta2 = beta3;
		gamma1 = gamma2; gamma2 = gamma3;
		sigma1 = sigma2; sigma2 = sigma3;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}

/* ------------------------------------------------------------------ IDR(s), lis_solver_idrs.c:523-783
 * The shadow space P is filled from MT19937 (init_by_array {0x123,0x234,0x345,0x456}, genrand_real1) exactly as the
 * reference does (:578-585), so the same P -- and with it the same iteration history -- comes out.  The generator
 * below is the published algorithm of Matsumoto & Nishimura (the reference bundles their mt19937ar.c). */
typedef struct { unsigned long mt[624]; int mti; } mt19937;

static void mt_seed(mt19937 *g, unsigned long s)
{
	g->mt[0] = s & 0xffffffffUL;
	for (g->mti = 1; g->mti < 624; g->mti++)
		g->mt[g->mti] = (1812433253UL * (g->mt[g->mti - 1] ^ (g->mt[g->mti - 1] >> 30)) + (unsigned long)g->mti) & 0xffffffffUL;
}

static void mt_seed_array(mt19937 *g, const unsigned long *key, int len)
{
	int i = 1, j = 0, k = 624 > len ? 624 : len;
	mt_seed(g, 19650218UL);
	for (; k; k--) {
		g->mt[i] = ((g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1664525UL)) + key[j] + (unsigned long)j) & 0xffffffffUL;
		i++; j++;
		if (i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
		if (j >= len) j = 0;
	}
	for (k = 623; k; k--) {
		g->mt[i] = ((g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1566083941UL)) - (unsigned long)i) & 0xffffffffUL;
		i++;
		if (i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
	}
	g->mt[0] = 0x80000000UL;
}

static unsigned long mt_next(mt19937 *g)
{
	static const unsigned long mag[2] = {0x0UL, 0x9908b0dfUL};
	unsigned long y;
	if (g->mti >= 624) {
		int kk;
		for (kk = 0; kk < 624 - 397; kk++) {
			y = (g->mt[kk] & 0x80000000UL) | (g->mt[kk + 1] & 0x7fffffffUL);
			g->mt[kk] = g->mt[kk + 397] ^ (y >> 1) ^ mag[y & 1UL];
		}
		for (; kk < 623; kk++) {
			y = (g->mt[kk] & 0x80000000UL) | (g->mt[kk + 1] & 0x7fffffffUL);
			g->mt[kk] = g->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag[y & 1UL];
		}
		y = (g->mt[623] & 0x80000000UL) | (g->mt[0] & 0x7fffffffUL);
		g->mt[623] = g->mt[396] ^ (y >> 1) ^ mag[y & 1UL];
		g->mti = 0;
	}
	y = g->mt[g->mti++];
	y ^= (y >> 11);
	y ^= (y << 7) & 0x9d2c5680UL;
	y ^= (y << 15) & 0xefc60000UL;
	y ^= (y >> 18);
	return y & 0xffffffffUL;
}

/* x = M \ m for the s x s system of IDR(s) (column-major, lis_array_solve, src/array/lis_array.c): the reference's
 * LU without pivoting with reciprocal pivots, special-cased for s = 1, 2 with its own operation order */
static void small_solve(int n, const double *a, const double *b, double *x, double *w)
{
	for (int i = 0; i < n * n; i++) w[i] = a[i];
	if (n == 1) { x[0] = b[0] / w[0]; return; }
	if (n == 2) {
		w[0] = 1.0 / w[0];
		w[1] *= w[0];
		w[3] -= w[1] * w[2];
		w[3] = 1.0 / w[3];
		x[0] = b[0];
		x[1] = b[1] - w[1] * x[0];
		x[1] *= w[3];
		x[0] -= w[2] * x[1];
		x[0] *= w[0];
		return;
	}
	for (int k = 0; k < n; k++) {
		w[k + k * n] = 1.0 / w[k + k * n];
		for (int i = k + 1; i < n; i++) {
			const double t = w[i + k * n] * w[k + k * n];
			for (int j = k + 1; j < n; j++) w[i + j * n] -= t * w[k + j * n];
			w[i + k * n] = t;
		}
	}
	for (int i = 0; i < n; i++) {
		x[i] = b[i];
		for (int j = 0; j < i; j++) x[i] -= w[i + j * n] * x[j];
	}
	for (int i = n - 1; i >= 0; i--) {
		for (int j = i + 1; j < n; j++) x[i] -= w[i + j * n] * x[j];
		x[i] *= w[i + i * n];
	}
}

LIS_INT lisk_idrs(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	/* IDR(1) (lis_idr1, lis_solver_idrs.c:222-520) is the same recurrence with s = 1 written out */
	const int n = c->n, sd = (s->options[LIS_OPTIONS_SOLVER] == LIS_SOLVER_IDR1) ? 1 : s->options[LIS_OPTIONS_IDRS_RESTART];
	double nrm2 = 0.0, om = 0.0, h;
	double *m = NULL, *cf = NULL, *M = NULL, *MM = NULL, *hostP = NULL, *coef = NULL;
	const double **vs = NULL;
	if (sd < 1 || sd > 40) { err = LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_IDRS_RESTART(=%D) must be in [1,40]\n", (LIS_INT)sd); goto done; }
	/* multi-rank jobs: like the reference under MPI (:578-584), every rank seeds the generator identically and fills
	 * its own n local entries of each shadow vector */
	m = (double *)calloc((size_t)sd, sizeof(double)); cf = (double *)calloc((size_t)sd, sizeof(double));
	M = (double *)calloc((size_t)sd * sd, sizeof(double)); MM = (double *)calloc((size_t)sd * sd, sizeof(double));
	coef = (double *)calloc((size_t)sd + 1, sizeof(double)); vs = (const double **)calloc((size_t)sd + 1, sizeof(double *));
	hostP = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
	if (!m || !cf || !M || !MM || !coef || !vs || !hostP) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)sd); goto done; }
	TRY(work_alloc(c, 4 + 3 * sd));
	double *r = c->work[0], *t = c->work[1], *v = c->work[2], *av = c->work[3];
	double **dX = &c->work[4], **P = &c->work[4 + sd], **dR = &c->work[4 + 2 * sd];
	int st = initial_residual(c, r);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	{
		mt19937 g;
		const unsigned long key[4] = {0x123, 0x234, 0x345, 0x456};
		mt_seed_array(&g, key, 4);
		for (int k = 0; k < sd; k++) {
			for (int i = 0; i < n; i++) hostP[i] = (double)mt_next(&g) * (1.0 / 4294967295.0);
			KTRY(liship_memcpy_h2d(P[k], hostP, sizeof(double) * (size_t)n, lisg.stream));
			KTRY(liship_stream_synchronize(lisg.stream));
		}
	}
	for (int j = 0; j < sd; j++) {                    /* lis_idrs_orth (:202-220): modified Gram-Schmidt on P */
		double rn, d;
		TRY(lisd_nrm2(n, P[j], &rn));
		SCALE(1.0 / rn, P[j]);
		for (int i = j + 1; i < sd; i++) { DOT(P[j], P[i], &d); AXPY(-d, P[j], P[i]); }
	}
	for (int k = 0; k < sd; k++) {                    /* s minimal-residual start-up steps */
		PSOLVE(r, dX[k]);
		MATVEC(dX[k], dR[k]);
		DOT(dR[k], dR[k], &h);
		DOT(dR[k], r, &om);
		om = om / h;
		SCALE(om, dX[k]);
		SCALE(-om, dR[k]);
		AXPY(1.0, dX[k], c->x);
		AXPY(1.0, dR[k], r);
		RESID(r, &nrm2);
		iter = k + 1;
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		for (int i = 0; i < sd; i++) DOT(P[i], dR[k], &M[k * sd + i]);
	}
	iter = sd;
	int oldest = 0;
	for (int i = 0; i < sd; i++) DOT(P[i], r, &m[i]);
	/* lis_idr1 runs its two kinds of step back to back and tests maxiter once per pair (:320-510) */
	const int idr1 = (s->options[LIS_OPTIONS_SOLVER] == LIS_SOLVER_IDR1);
	while (iter <= c->maxiter || (idr1 && iter % 2 == 0)) {
		small_solve(sd, M, m, cf, MM);
		COPY(r, v);
		for (int j = 0; j < sd; j++) AXPY(-cf[j], dR[j], v);
		PSOLVE(v, av);
		for (int j = 0; j < sd; j++) { coef[j + 1] = -cf[j]; }
		if ((iter % (sd + 1)) == sd) {
			MATVEC(av, t);
			DOT(t, t, &h);
			DOT(t, v, &om);
			om = om / h;
			coef[0] = om;  vs[0] = av; for (int j = 0; j < sd; j++) vs[j + 1] = dX[j];     /* dX[oldest] = om av - sum c_j dX_j */
			KTRY(liship_lincomb_f64(n, sd + 1, vs, coef, 0, dX[oldest], lisg.stream));
			coef[0] = -om; vs[0] = t;  for (int j = 0; j < sd; j++) vs[j + 1] = dR[j];     /* dR[oldest] = -om t - sum c_j dR_j */
			KTRY(liship_lincomb_f64(n, sd + 1, vs, coef, 0, dR[oldest], lisg.stream));
		} else {
			coef[0] = om;  vs[0] = av; for (int j = 0; j < sd; j++) vs[j + 1] = dX[j];
			KTRY(liship_lincomb_f64(n, sd + 1, vs, coef, 0, dX[oldest], lisg.stream));
			MATVEC(dX[oldest], dR[oldest]);
			SCALE(-1.0, dR[oldest]);
		}
		AXPY(1.0, dR[oldest], r);
		AXPY(1.0, dX[oldest], c->x);
		iter++;
		RESID(r, &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) FINISH(LIS_SUCCESS);
		for (int i = 0; i < sd; i++) {
			DOT(P[i], dR[oldest], &h);
			m[i] += h;
			M[oldest * sd + i] = h;
		}
		oldest++;
		if (oldest == sd) oldest = 0;
	}
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	free(m); free(cf); free(M); free(MM); free(coef); free((void *)vs); free(hostP);
	return err;
}

/* ------------------------------------------------------------------ BiCGSTAB(l), lis_solver_bicgstabl.c:136-419
 * l BiCG steps building r[0..l], u[0..l] with the right-preconditioned operator A M^-1, then a degree-l minimal
 * residual polynomial (modified Gram-Schmidt on r[1..l], small triangular solves on the host).  The iterate is
 * kept in preconditioned space and mapped back on exit as x = M^-1 x + x0 -- including the reference's handling
 * of a non-zero initial guess and of LIS_MAXITER (no map back) -- (:183-184, :278-281, :407-414). */
LIS_INT lisk_bicgstabl(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n, l = s->options[LIS_OPTIONS_ELL], zd = l + 1;
	double nrm2 = 0.0, alpha = 0.0, beta, omega = 1.0, rho0 = 1.0, rho1, nu, rnorm0, rnorm, normx, normr;
	double *tau = NULL;
	if (l < 1 || l > 30) { err = LISI_ERR(LIS_ERR_ILL_ARG, "Parameter LIS_OPTIONS_ELL(=%D) must be in [1,30]\n", (LIS_INT)l); goto done; }
	tau = (double *)calloc((size_t)zd * (size_t)(4 + l + 1), sizeof(double));
	if (!tau) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)zd); goto done; }
	double *gamma = tau + zd * zd, *gamma1 = gamma + zd, *gamma2 = gamma1 + zd, *sigma = gamma2 + zd;
	TRY(work_alloc(c, 4 + 2 * (l + 1)));
	double *rtld = c->work[0], *xp = c->work[1], *bp = c->work[2], *t = c->work[3], **r = &c->work[4], **u = &c->work[l + 1 + 4];
	int st = initial_residual(c, r[0]);
	if (st) { err = st < 0 ? -st : 0; goto done; }
	COPY(r[0], rtld);
	COPY(r[0], bp);
	COPY(c->x, xp);
	TRY(lisd_nrm2(n, r[0], &rnorm0));
	rnorm = normx = normr = rnorm0;
	(void)rnorm; (void)normx; (void)normr;
#define MAP_BACK() do { PSOLVE(c->x, t); COPY(t, c->x); AXPY(1.0, xp, c->x); } while (0)
	while (iter <= c->maxiter) {
		rho0 = -omega * rho0;
		for (int j = 0; j < l; j++) {                 /* BiCG part */
			iter++;
			DOT(rtld, r[j], &rho1);
			if (rho1 == 0.0) { MAP_BACK(); FINISH(LIS_BREAKDOWN); }
			beta = alpha * (rho1 / rho0);
			rho0 = rho1;
			for (int i = 0; i <= j; i++) XPAY(r[i], -beta, u[i]);
			PSOLVE(u[j], t);
			MATVEC(t, u[j + 1]);
			DOT(rtld, u[j + 1], &nu);
			if (nu == 0.0) { MAP_BACK(); FINISH(LIS_BREAKDOWN); }
			alpha = rho1 / nu;
			AXPY(alpha, u[0], c->x);
			for (int i = 0; i <= j; i++) AXPY(-alpha, u[i + 1], r[i]);
			RESID(r[0], &nrm2);
			if (iter % l != 0) note(c, iter, nrm2);
			if (c->tol >= nrm2) {
				if (iter % l == 0) note(c, iter, nrm2);
				MAP_BACK();
				FINISH(LIS_SUCCESS);
			}
			PSOLVE(r[j], t);
			MATVEC(t, r[j + 1]);
			TRY(lisd_nrm2(n, r[0], &rnorm));            /* kept: the reference tracks max norms here (unused) */
		}
		for (int j = 1; j <= l; j++) {                /* MR part */
			for (int i = 1; i <= j - 1; i++) {
				DOT(r[j], r[i], &nu);
				nu = nu / sigma[i];
				tau[i * zd + j] = nu;
				AXPY(-nu, r[i], r[j]);
			}
			DOT(r[j], r[j], &sigma[j]);
			DOT(r[0], r[j], &nu);
			gamma1[j] = nu / sigma[j];
		}
		gamma[l] = gamma1[l];
		omega = gamma[l];
		for (int j = l - 1; j >= 1; j--) {
			nu = 0.0;
			for (int i = j + 1; i <= l; i++) nu += tau[j * zd + i] * gamma[i];
			gamma[j] = gamma1[j] - nu;
		}
		for (int j = 1; j <= l - 1; j++) {
			nu = 0.0;
			for (int i = j + 1; i <= l - 1; i++) nu += tau[j * zd + i] * gamma[i + 1];
			gamma2[j] = gamma[j + 1] + nu;
		}
		AXPY(gamma[1], r[0], c->x);                   /* update */
		AXPY(-gamma1[l], r[l], r[0]);
		AXPY(-gamma[l], u[l], u[0]);
		for (int j = 1; j <= l - 1; j++) {
			AXPY(-gamma[j], u[j], u[0]);
			AXPY(gamma2[j], r[j], c->x);
			AXPY(-gamma1[j], r[j], r[0]);
		}
		RESID(r[0], &nrm2);
		note(c, iter, nrm2);
		if (c->tol >= nrm2) { MAP_BACK(); FINISH(LIS_SUCCESS); }
	}
	FINISH(LIS_MAXITER);
#undef MAP_BACK
done:
	work_free(c);
	free(tau);
	return err;
}

/* the stationary Jacobi iteration as a solver, lis_solver_jacobi.c:104-190 (-p none; with a preconditioner the
 * reference first rescales the system, lis_solver.c:639-656: not served).  x += D^-1 (b - A x), residual relative
 * to ||b||, against the raw -tol parameter. */
LIS_INT lisk_jacobi(ctx_t *c)
{
	LIS_SOLVER s = c->s;
	LIS_INT err = 0, iter = 0;
	const int n = c->n;
	const double tol = s->params[LIS_PARAMS_RESID - LIS_OPTIONS_LEN];
	double nrm2 = 0.0, bnrm2;
	lisd_mat *dm = MDEV(c->A);
	if (c->dinv) { err = LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "the Jacobi solver with a preconditioner (system rescaling) is not served by liblis_amd\n"); goto done; }
	if (dm->type != LIS_MATRIX_CSR) { err = LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "the Jacobi solver is served for CSR / CSC storage only\n"); goto done; }
	TRY(work_alloc(c, 4));
	double *r = c->work[0], *t = c->work[1], *sx = c->work[2], *d = c->work[3];
	TRY(lisd_nrm2(n, c->b, &bnrm2));
	bnrm2 = 1.0 / bnrm2;
	KTRY(liship_csr_diagonal_f64(n, dm->ptr, dm->index, dm->value, d, lisg.stream));
	KTRY(liship_reciprocal_f64(n, d, lisg.stream));
	for (iter = 1; iter <= c->maxiter; iter++) {
		PSOLVE(c->x, sx);
		MATVEC(sx, t);
		AXPYZ(-1.0, t, c->b, r);
		TRY(lisd_nrm2(n, r, &nrm2));
		KTRY(liship_pmul_f64(n, r, d, r, lisg.stream));
		AXPY(1.0, r, c->x);
		nrm2 = nrm2 * bnrm2;
		note(c, iter, nrm2);
		if (tol >= nrm2) { PSOLVE(c->x, sx); COPY(sx, c->x); FINISH(LIS_SUCCESS); }
	}
	PSOLVE(c->x, sx); COPY(sx, c->x);
	FINISH(LIS_MAXITER);
done:
	work_free(c);
	return err;
}
