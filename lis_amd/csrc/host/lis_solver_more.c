/*
 * lis_solver_more.c -- the other short-recurrence Krylov solvers of Lis on the same HBM work vectors and the
 * same kernels: CGS, CR, GPBiCG, TFQMR, BiCGSafe, Orthomin(m).
 *
 * Each loop issues the reference's vector operations in the reference's order, one kernel per call
 * (element-wise results are bit-identical, reductions are the deterministic trees of vector_ops.hip), so the
 * recurrences see the same numbers as the CPU path up to the reduction order.  No pass fusion here: these are
 * coverage, CG / BiCG / BiCGSTAB / GMRES in lis_solver.c are the tuned ones.
 *   CGS       src/solver/lis_solver_cgs.c:128-262        CR         lis_solver_cg.c:681-795
 *   GPBiCG    lis_solver_gpbicg.c:145-351                TFQMR      lis_solver_qmr.c:113-300
 *   BiCGSafe  lis_solver_bicgsafe.c:145-326              Orthomin   lis_solver_orthomin.c:124-252
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
	       *qhat = c->work[5], *u = c->work[5], *uhat = c->work[6], *vhat = c->work[6];      /* aliases as :147-153 */
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

LIS_INT lisk_gpbicg(ctx_t *c)
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
	COPY(r, rtld);
	PSOLVE(r, p);
	DOT(rtld, r, &rho_old);
	for (iter = 1; iter <= c->maxiter; iter++) {
		MATVEC(p, ap);
		PSOLVE(ap, map);
		DOT(rtld, ap, &t5[0]);
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
		DOT(rtld, r, &rho);
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
