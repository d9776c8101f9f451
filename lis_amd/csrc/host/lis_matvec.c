/*
 * lis_matvec.c -- y = A x through the Lis entry points, executed by the HIP kernels.
 *
 *   lis_matvec(A, X, Y)               reference src/matvec/lis_matvec.c:55-187: switch on A->matrix_type,
 *                                     halo exchange first in a multi-rank job (LIS_MATVEC_SENDRECV,
 *                                     include/lis_matvec.h:31-44), unknown format -> LIS_ERR_NOT_IMPLEMENTED
 *   lis_matvec_<fmt>(A, x[], y[])     reference include/lis_matvec.h:91-181: raw HOST arrays.  They cannot
 *                                     report errors (void), so a failing device call aborts loudly.
 */
#include <stdio.h>
#include "lis_internal.h"

LIS_INT lis_matvec(LIS_MATRIX A, LIS_VECTOR X, LIS_VECTOR Y)
{
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR: case LIS_MATRIX_CSC: case LIS_MATRIX_ELL:
	case LIS_MATRIX_DIA: case LIS_MATRIX_JAD: case LIS_MATRIX_BSR:
		break;
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
	LISCHK(lisd_mat_ready(A));
	/* like the reference (LIS_MATVEC_REALLOC, include/lis_matvec.h:32-43: lis_realloc of X->value to np + pad entries), X grows to hold the ghost and block
	 * padding entries of A -- the host array too: a program that reads X->value[n ..) after a product in a multi-rank job must find memory there.  The new
	 * entries are zero, the old ones keep their place, nothing is transferred: whichever copy is current stays the truth. */
	if (A->np + A->pad > X->np + X->pad) {
		const size_t need = (size_t)A->np + (size_t)A->pad;
		lisd_vec *dv = VDEV(X);
		if (X->value && dv->hlen < need) {
			const int hv = dv->host_valid, dvv = dv->dev_valid;
			LISCHK(lisp_grow(X, need));
			if (lisp_lazy() && dv->region) lisp_protect(X, (!hv && dvv) ? LISP_NONE : (hv && dvv) ? LISP_RO : LISP_RW);
		}
		X->np = A->np; X->pad = A->pad;
	}
	double *dx, *dy;
	size_t need_x = (size_t)(A->np + A->pad), need_y = (size_t)(A->n + A->pad);
	if (A->matrix_type == LIS_MATRIX_BSR) { need_x = (size_t)A->nc * A->bnc; need_y = (size_t)A->nr * A->bnr; }
	LISCHK(lisd_vec_reserve(X, need_x));
	LISCHK(lisd_vec_reserve(Y, need_y));
	LISCHK(lisd_vec_in(X, &dx));
	LISCHK(lisd_vec_out(Y, &dy));
	LISCHK(lisd_spmv(A, dx, dy));
	return lisd_vec_done(Y);
}

static void raw_matvec(LIS_MATRIX A, LIS_INT fmt, LIS_SCALAR x[], LIS_SCALAR y[])
{
	lisd_mat *d = MDEV(A);
	LIS_INT err = (A->matrix_type == fmt) ? lisd_mat_ready(A) : LIS_ERR_ILL_ARG;
	const size_t nx = (size_t)A->np + (size_t)A->pad + 16 + (fmt == LIS_MATRIX_BSR ? (size_t)A->nc * A->bnc : 0);
	if (!err && d->scap < nx) {
		(void)liship_free(d->sx); (void)liship_free(d->sy);
		d->sx = d->sy = NULL; d->scap = 0;
		if (lisd_malloc((void **)&d->sx, nx * sizeof(double)) || lisd_malloc((void **)&d->sy, nx * sizeof(double))) err = LIS_ERR_OUT_OF_MEMORY;
		else { d->scap = nx; (void)liship_memset(d->sx, 0, nx * sizeof(double), lisg.stream); }
	}
	if (!err) { int rc = liship_memcpy_h2d(d->sx, x, sizeof(double) * (size_t)(lisg.nprocs > 1 ? A->n : A->np), lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	if (!err) err = lisd_spmv(A, d->sx, d->sy);
	if (!err) { int rc = liship_memcpy_d2h(y, d->sy, sizeof(double) * (size_t)A->n, lisg.stream); if (!rc) rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	if (err) {
		fprintf(stderr, "liblis_amd: lis_matvec_<fmt>(A, x[], y[]) failed (code %d) and has no error channel -- aborting\n", (int)err);
		abort();
	}
}

void lis_matvec_csr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_CSR, x, y); }
void lis_matvec_csc(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_CSC, x, y); }
void lis_matvec_ell(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_ELL, x, y); }
void lis_matvec_dia(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_DIA, x, y); }
void lis_matvec_jad(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_JAD, x, y); }
void lis_matvec_bsr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvec(A, LIS_MATRIX_BSR, x, y); }

/* ref src/matvec/lis_matvec.c:50-51 */
LIS_MATVEC_FUNC LIS_MATVEC  = lis_matvec;
LIS_MATVEC_FUNC LIS_MATVECH = lis_matvech;

/* ref src/matvec/lis_matvec.c:354-461: the same sweep and the same report lines over the formats this library serves (the reference walks all eleven and CHKERRs on a
 * conversion it cannot do; MSR, BSC, VBR, COO and DNS are reported as not served here).  Products are asynchronous: the clock stops behind ONE synchronize per format,
 * so `computation` is device time for `iter` products, not `iter` launch times. */
LIS_INT lis_matvec_optimize(LIS_MATRIX A, LIS_INT *matrix_type_maxperf)
{
	static const char *name[] = {"CSR", "CSC", "MSR", "DIA", "ELL", "JAD", "BSR", "BSC", "VBR", "COO", "DNS"};
	if (!A || !matrix_type_maxperf) return LISI_ERR(LIS_ERR_ILL_ARG, "lis_matvec_optimize: NULL argument\n");
	LIS_VECTOR X = NULL, Y = NULL;
	LIS_INT err = lis_vector_duplicate(A, &X);
	if (!err) err = lis_vector_duplicate(A, &Y);
	if (!err) err = lis_vector_set_all(1.0, X);
	if (err) { if (X) lis_vector_destroy(X); if (Y) lis_vector_destroy(Y); return err; }
	const LIS_INT iter = (LIS_INT)(10000000 / (A->nnz > 0 ? A->nnz : 1)) + 1;
	double best = 0.0;
	*matrix_type_maxperf = A->matrix_type;
	if (lisg.rank == 0) {
		printf("\nmeasuring matvec performance...\n");
		printf("number of iterations = 1e7 / %d + 1 = %d\n", (int)A->nnz, (int)iter);
	}
	for (LIS_INT type = 1; type < 11 && !err; type++) {
		const int served = type == LIS_MATRIX_CSR || type == LIS_MATRIX_CSC || type == LIS_MATRIX_DIA || type == LIS_MATRIX_ELL || type == LIS_MATRIX_JAD || type == LIS_MATRIX_BSR;
		if (!served) {
			if (lisg.rank == 0) printf("matrix_type = %2d (%s), not served by liblis_amd\n", (int)type, name[type - 1]);
			continue;
		}
		LIS_MATRIX A1 = NULL;
		err = lis_matrix_duplicate(A, &A1);
		if (!err) err = lis_matrix_set_type(A1, type);
		if (!err) err = lis_matrix_convert(A, A1);
		if (!err) err = lis_matvec(A1, X, Y);                     /* untimed: uploads A1 and builds its plan (the reference converts outside its clock too) */
		if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
		double comptime = lis_wtime();
		for (LIS_INT i = 0; i < iter && !err; i++) err = lis_matvec(A1, X, Y);
		if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
		comptime = lis_wtime() - comptime;
		LIS_REAL val = 0.0;
		if (!err) err = lis_vector_nrm2(Y, &val);
		if (!err) {
			const double flops = comptime > 0.0 ? 2.0 * (double)A->nnz * (double)iter * 1.0e-6 / comptime : 0.0;
			if (lisg.rank == 0) printf("matrix_type = %2d (%s), computation = %e sec, %8.3f MFLOPS\n", (int)type, name[type - 1], comptime, flops);
			if (flops > best) { best = flops; *matrix_type_maxperf = type; }
		}
		if (A1) lis_matrix_destroy(A1);
	}
	if (!err && lisg.rank == 0) printf("matrix format is set to %s\n\n", name[*matrix_type_maxperf - 1]);
	fflush(stdout);
	lis_vector_destroy(X); lis_vector_destroy(Y);
	return err;
}
