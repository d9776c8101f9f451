/*
 * lis_matvech.c -- y = A^T x (A^H for the real build) through lis_matvech, executed by the CSR kernel.
 *
 * The reference computes A^T x by SCATTER: it walks the stored matrix in its native order and does
 * y[col] += value * x[row] (src/matvec/lis_matvec_csr.c:213-250, _ell.c:181-215, _dia.c:262-310,
 * _jad.c:545-580, _bsr.c:935-957; CSC is a plain row sum, _csc.c:176-190).  A scatter wants atomics on a
 * GPU and would make the sum order a race.  Instead the transposed operator is materialised ONCE per
 * matrix as a CSR whose row c lists the contributions to y[c] in exactly the order the reference's
 * walk produces them (a stable counting sort of the walk by destination).  Then
 *     y[c] = sum of that row, left to right from +0.0
 * is bit-identical to the reference at one thread, and A^T x runs at the speed of the row-gather CSR
 * kernel (spmv_csr.hip) instead of an atomic scatter.  Cost: a second copy of the matrix in HBM
 * (288 GB per GPU: affordable), built lazily on the first lis_matvech.
 * Multi-GPU: local rows of A^T are the np columns of the local block, so y has ghost entries; they are
 * sent back to their owners and added there in neighbour order (lis_reduce, lis_matrix_mpi.c:959-1000).
 *
 * Split matrices (A = L + D + U, lis_split.c), the two formats whose split A^T x is reached by solver options (`-scale jacobi -storage bsr`
 * with BiCG-type solvers, and lis_matrix_split by hand on CSR):
 *   CSR  the OpenMP build scatters the L entries, then the U entries of row i into w, row after row, and finishes with
 *        y[c] = D[c]*x[c] + (0.0 + w[c]) (src/matvec/lis_matvec_csr.c:125-160): the transposed rows list the off-diagonal walk, the
 *        product forms w, one element-wise pass adds it to D.*x -- (d x) + (sum), not one chain
 *   BSR  serial in every build (lis_matvec_bsr.c:878-925): y = 0, then the diagonal blocks' terms, then the L and U blocks of block
 *        row after block row: ONE chain per y[c] -- the transposed rows list exactly that walk, D terms first
 */
#include <stdio.h>
#include "lis_internal.h"

typedef struct {
	int pass;                 /* 0: count, 1: place */
	LIS_INT rows, cols;       /* destinations [0,rows), sources [0,cols) */
	LIS_INT *ptr, *fill, *index;
	double *value;
} walk_t;

static inline void emit(walk_t *w, LIS_INT dst, LIS_INT src, double v)
{
	if (dst >= w->rows || src >= w->cols) return;        /* block padding of BSR */
	if (w->pass == 0) w->ptr[dst + 1]++;
	else { const LIS_INT at = w->fill[dst]++; w->index[at] = src; w->value[at] = v; }
}

/* the reference's scatter order, one thread */
static void walk(LIS_INT type, LIS_MATRIX A, const LIS_INT *ptr, const LIS_INT *index, const double *value, walk_t *w)
{
	const LIS_INT n = A->n;
	switch (type) {
	case LIS_MATRIX_CSR:
		for (LIS_INT i = 0; i < n; i++)
			for (LIS_INT j = ptr[i]; j < ptr[i + 1]; j++) emit(w, index[j], i, value[j]);
		break;
	case LIS_MATRIX_ELL:
		for (LIS_INT j = 0; j < A->maxnzr; j++)
			for (LIS_INT i = 0; i < n; i++) emit(w, index[(size_t)j * n + i], i, value[(size_t)j * n + i]);
		break;
	case LIS_MATRIX_DIA:
		for (LIS_INT d = 0; d < A->nnd; d++) {
			const LIS_INT jj = index[d];
			const LIS_INT lo = jj < 0 ? -jj : 0;
			const LIS_INT hi = (lisg.nprocs > 1) ? (jj <= A->np - n ? n : (A->np - jj < n ? A->np - jj : n)) : (n - jj < n ? n - jj : n);
			for (LIS_INT i = lo; i < hi; i++) emit(w, jj + i, i, value[(size_t)d * n + i]);
		}
		break;
	case LIS_MATRIX_JAD:
		for (LIS_INT j = 0; j < A->maxnzr; j++) {
			LIS_INT k = 0;
			for (LIS_INT i = ptr[j]; i < ptr[j + 1]; i++, k++) emit(w, index[i], A->row[k], value[i]);
		}
		break;
	case LIS_MATRIX_BSR: {
		const LIS_INT bnr = A->bnr, bnc = A->bnc, bs = bnr * bnc;
		for (LIS_INT bi = 0; bi < A->nr; bi++)
			for (LIS_INT bc = A->bptr[bi]; bc < A->bptr[bi + 1]; bc++) {
				const LIS_INT bj = A->bindex[bc] * bnc;
				size_t k = (size_t)bc * bs;
				for (LIS_INT j = 0; j < bnc; j++)
					for (LIS_INT i = 0; i < bnr; i++, k++) emit(w, bj + j, bi * bnr + i, value[k]);
			}
		break; }
	default: break;
	}
}

/* the walks of a split matrix (see the head of this file); CSR: off-diagonal parts only, the diagonal is added by lisd_spmv_t */
static void walk_split(LIS_MATRIX A, walk_t *w)
{
	const LIS_INT n = A->n;
	if (A->matrix_type == LIS_MATRIX_CSR) {
		for (LIS_INT i = 0; i < n; i++) {
			for (LIS_INT j = A->L->ptr[i]; j < A->L->ptr[i + 1]; j++) emit(w, A->L->index[j], i, A->L->value[j]);
			for (LIS_INT j = A->U->ptr[i]; j < A->U->ptr[i + 1]; j++) emit(w, A->U->index[j], i, A->U->value[j]);
		}
		return;
	}
	const LIS_INT bnr = A->bnr, bnc = A->bnc, bs = bnr * bnc;
	for (LIS_INT bi = 0; bi < A->nr; bi++) {
		size_t k = (size_t)bi * bs;
		for (LIS_INT j = 0; j < bnc; j++)
			for (LIS_INT i = 0; i < bnr; i++, k++) emit(w, bi * bnr + j, bi * bnr + i, A->D->value[k]);
	}
	for (LIS_INT bi = 0; bi < A->nr; bi++)
		for (int part = 0; part < 2; part++) {
			const LIS_MATRIX_CORE P = part ? A->U : A->L;
			for (LIS_INT bc = P->bptr[bi]; bc < P->bptr[bi + 1]; bc++) {
				const LIS_INT bj = P->bindex[bc] * bnc;
				size_t k = (size_t)bc * bs;
				for (LIS_INT j = 0; j < bnc; j++)
					for (LIS_INT i = 0; i < bnr; i++, k++) emit(w, bj + j, bi * bnr + i, P->value[k]);
			}
		}
}

static LIS_INT upload(void **dst, const void *src, size_t bytes)
{
	HIPCHK(lisd_malloc(dst, bytes + 16));
	if (bytes) HIPCHK(liship_memcpy_h2d(*dst, src, bytes, lisg.stream));
	return LIS_SUCCESS;
}

LIS_INT lisd_mat_ready_t(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	const int split = A->is_splited != 0;
	if (split && !(A->matrix_type == LIS_MATRIX_CSR || (A->matrix_type == LIS_MATRIX_BSR && A->bnr == A->bnc)))
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "A^T x of a split (D/L/U) matrix is served for CSR and square-block BSR storage\n");
	/* (a multi-rank job: the transposed rows of the split walk cover the np local columns like any other matrix's, the ghost rows' sums travel back to their owners
	 * through the reverse exchange -- lisd_spmv_t / lisc_reduce_device, the reference's LIS_MATVEC_REDUCE around lis_matvech_<fmt>, src/matvec/lis_matvec.c:191-349) */
	if (d->t_ready) return LIS_SUCCESS;
	LISCHK(lisd_mat_ready(A));
	if (!(A->matrix_type == LIS_MATRIX_CSR && d->type == LIS_MATRIX_CSR)) LISCHK(lisp_fill_matrix(A));   /* the host arrays are read below (and handed to the runtime) */
	const LIS_INT type = A->matrix_type, n = A->n;
	/* rows of A^T = local columns.  BSR in a multi-rank job: the ghost block columns start on a fresh block column, i.e. pad entries behind the owned ones
	 * (lis_matrix_bsr.c:425-428; the halo lands at x[n + pad ...), lis_comm.c), so the ghost rows of A^T reach np + pad -- with np alone the walk dropped the
	 * contributions to the last `pad` ghosts (round 6: found by the 3 x 3 blocks of the split multi-rank test; square 2 x 2 blocks of an even slab have pad 0) */
	const LIS_INT np = (lisg.nprocs > 1 && type == LIS_MATRIX_BSR) ? A->np + A->pad : A->np;
	d->t_rows = np;
	if (split) {
		walk_t w;
		memset(&w, 0, sizeof(w));
		w.rows = np; w.cols = n;
		w.ptr = (LIS_INT *)calloc((size_t)np + 2, sizeof(LIS_INT));
		if (!w.ptr) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "transpose\n");
		w.pass = 0;
		walk_split(A, &w);
		for (LIS_INT c = 0; c < np; c++) w.ptr[c + 1] += w.ptr[c];
		const LIS_INT tnnz = w.ptr[np];
		w.fill = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(np > 0 ? np : 1));
		w.index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(tnnz > 0 ? tnnz : 1));
		w.value = (double *)malloc(sizeof(double) * (size_t)(tnnz > 0 ? tnnz : 1));
		LIS_INT err = LIS_SUCCESS;
		if (!w.fill || !w.index || !w.value) err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "transpose\n");
		if (!err) {
			memcpy(w.fill, w.ptr, sizeof(LIS_INT) * (size_t)np);
			w.pass = 1;
			walk_split(A, &w);
			d->t_nnz = tnnz;
			err = upload((void **)&d->t_ptr, w.ptr, sizeof(int) * ((size_t)np + 1));
			if (!err) err = upload((void **)&d->t_index, w.index, sizeof(int) * (size_t)tnnz);
			if (!err) err = upload((void **)&d->t_value, w.value, sizeof(double) * (size_t)tnnz);
			if (!err && type == LIS_MATRIX_CSR) err = upload((void **)&d->t_diag, A->D->value, sizeof(double) * (size_t)n);
			if (!err) { int rc__ = liship_stream_synchronize(lisg.stream); if (rc__) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc__); }
		}
		free(w.ptr); free(w.fill); free(w.index); free(w.value);
		if (err) return err;
	} else if (type == LIS_MATRIX_CSC) {          /* CSC arrays ARE the CSR of A^T: np rows, row indices as columns */
		d->t_nnz = A->nnz;
		LISCHK(upload((void **)&d->t_ptr, A->ptr, sizeof(int) * ((size_t)np + 1)));
		LISCHK(upload((void **)&d->t_index, A->index, sizeof(int) * (size_t)A->nnz));
		LISCHK(upload((void **)&d->t_value, A->value, sizeof(double) * (size_t)A->nnz));
	} else if (type == LIS_MATRIX_CSR && d->type == LIS_MATRIX_CSR) {
		/* CSR: the HBM copy is transposed in HBM (transpose.hip): no host pass, works for matrices born on the device */
		int *work = NULL;
		d->t_nnz = d->nnz;
		HIPCHK(lisd_malloc((void **)&d->t_ptr, sizeof(int) * ((size_t)np + 1) + 16));
		HIPCHK(lisd_malloc((void **)&d->t_index, sizeof(int) * (size_t)d->nnz + 16));
		HIPCHK(lisd_malloc((void **)&d->t_value, sizeof(double) * (size_t)d->nnz + 16));
		HIPCHK(lisd_malloc((void **)&work, sizeof(int) * ((size_t)np + (size_t)d->nnz) + 16));
		int rc = liship_csr_transpose_f64(n, np, d->nnz, d->ptr, d->index, d->value, d->t_ptr, d->t_index, d->t_value, work, lisg.stream);
		if (!rc) rc = liship_stream_synchronize(lisg.stream);
		(void)liship_free(work);
		HIPCHK(rc);
	} else {
		const LIS_INT *ptr = A->ptr, *index = A->index;
		const double *value = A->value;
		LIS_INT *hptr = NULL, *hidx = NULL;
		double *hval = NULL;
		if (d->device_only) {                /* matrix was born in HBM: bring the arrays down for the one-off sort */
			hptr = (LIS_INT *)malloc(sizeof(LIS_INT) * ((size_t)n + 1));
			hidx = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(d->nnz > 0 ? d->nnz : 1));
			hval = (double *)malloc(sizeof(double) * (size_t)(d->nnz > 0 ? d->nnz : 1));
			if (!hptr || !hidx || !hval) { free(hptr); free(hidx); free(hval); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "transpose staging\n"); }
			HIPCHK(liship_memcpy_d2h(hptr, d->ptr, sizeof(LIS_INT) * ((size_t)n + 1), lisg.stream));
			HIPCHK(liship_memcpy_d2h(hidx, d->index, sizeof(LIS_INT) * (size_t)d->nnz, lisg.stream));
			HIPCHK(liship_memcpy_d2h(hval, d->value, sizeof(double) * (size_t)d->nnz, lisg.stream));
			HIPCHK(liship_stream_synchronize(lisg.stream));
			ptr = hptr; index = hidx; value = hval;
		}
		walk_t w;
		memset(&w, 0, sizeof(w));
		w.rows = np; w.cols = n;
		w.ptr = (LIS_INT *)calloc((size_t)np + 2, sizeof(LIS_INT));
		if (!w.ptr) { free(hptr); free(hidx); free(hval); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "transpose\n"); }
		w.pass = 0;
		walk(type, A, ptr, index, value, &w);
		for (LIS_INT c = 0; c < np; c++) w.ptr[c + 1] += w.ptr[c];
		const LIS_INT tnnz = w.ptr[np];
		w.fill = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(np > 0 ? np : 1));
		w.index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(tnnz > 0 ? tnnz : 1));
		w.value = (double *)malloc(sizeof(double) * (size_t)(tnnz > 0 ? tnnz : 1));
		LIS_INT err = LIS_SUCCESS;
		if (!w.fill || !w.index || !w.value) err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "transpose\n");
		if (!err) {
			memcpy(w.fill, w.ptr, sizeof(LIS_INT) * (size_t)np);
			w.pass = 1;
			walk(type, A, ptr, index, value, &w);
			d->t_nnz = tnnz;
			err = upload((void **)&d->t_ptr, w.ptr, sizeof(int) * ((size_t)np + 1));
			if (!err) err = upload((void **)&d->t_index, w.index, sizeof(int) * (size_t)tnnz);
			if (!err) err = upload((void **)&d->t_value, w.value, sizeof(double) * (size_t)tnnz);
			if (!err) { int rc__ = liship_stream_synchronize(lisg.stream); if (rc__) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc__); }
		}
		free(w.ptr); free(w.fill); free(w.index); free(w.value);
		free(hptr); free(hidx); free(hval);
		if (err) return err;
	}
	LISCHK(lisd_csr_plan_plain(&d->t_plan, d->t_rows, d->t_ptr, d->t_index, d->t_value));
	HIPCHK(liship_stream_synchronize(lisg.stream));
	d->t_ready = 1;
	return LIS_SUCCESS;
}

/* y[0..np) = A^T x on device pointers, ghost contributions folded back to their owners */
LIS_INT lisd_spmv_t(LIS_MATRIX A, double *dx, double *dy)
{
	lisd_mat *d = MDEV(A);
	LISCHK(lisd_mat_ready_t(A));
	if (lisg.ref_reductions > 1 && A->matrix_type == LIS_MATRIX_CSR)
		/* reference-order mode at T > 1: the OpenMP build's per-thread scatter buffers group a column's terms by row chunk (lis_matvec_csr.c:207-236).
		 * CSR only: the other formats' threaded walks group differently again and keep the one-thread order */
		HIPCHK(liship_spmv_csr_transposed_chunked_f64(d->t_rows, A->n, lisg.ref_reductions, d->t_ptr, d->t_index, d->t_value, dx, dy, lisg.stream));
	else
		HIPCHK(liship_spmv_csr_f64(d->t_plan, d->t_ptr, d->t_index, d->t_value, dx, dy, lisg.stream));
	if (d->t_diag)          /* split CSR: y = D.*x + 1*w, w the off-diagonal sums just formed (1*w is w; lis_matvec_csr.c:152-159) */
		HIPCHK(liship_pmul_xpay_f64(A->n, dx, d->t_diag, 1.0, dy, lisg.stream));
	if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_reduce_device(A, dy));
	return LIS_SUCCESS;
}

static LIS_INT served(LIS_MATRIX A)
{
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR: case LIS_MATRIX_CSC: case LIS_MATRIX_ELL:
	case LIS_MATRIX_DIA: case LIS_MATRIX_JAD: case LIS_MATRIX_BSR:
		return LIS_SUCCESS;
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
}

LIS_INT lis_matvech(LIS_MATRIX A, LIS_VECTOR X, LIS_VECTOR Y)
{	/* ref src/matvec/lis_matvec.c:191-349; Y grows to np+pad like LIS_MATVEC_REDUCE0 (lis_matvec.h:60-72) */
	LISCHK(served(A));
	LISCHK(lisd_mat_ready_t(A));
	if (A->np + A->pad > Y->np + Y->pad) { Y->np = A->np; Y->pad = A->pad; }
	double *dx, *dy;
	LISCHK(lisd_vec_reserve(X, (size_t)(A->np + A->pad)));
	LISCHK(lisd_vec_reserve(Y, (size_t)(A->np + A->pad)));
	LISCHK(lisd_vec_in(X, &dx));
	LISCHK(lisd_vec_out(Y, &dy));
	LISCHK(lisd_spmv_t(A, dx, dy));
	return lisd_vec_done(Y);
}

/* raw HOST arrays, ref include/lis_matvec.h:92-178; void: a failing device call aborts loudly */
static void raw_matvech(LIS_MATRIX A, LIS_INT fmt, LIS_SCALAR x[], LIS_SCALAR y[])
{
	lisd_mat *d = MDEV(A);
	LIS_INT err = (A->matrix_type == fmt) ? lisd_mat_ready_t(A) : LIS_ERR_ILL_ARG;
	const size_t nx = (size_t)A->np + (size_t)A->pad + 16;
	if (!err && d->scap < nx) {
		(void)liship_free(d->sx); (void)liship_free(d->sy);
		d->sx = d->sy = NULL; d->scap = 0;
		if (lisd_malloc((void **)&d->sx, nx * sizeof(double)) || lisd_malloc((void **)&d->sy, nx * sizeof(double))) err = LIS_ERR_OUT_OF_MEMORY;
		else { d->scap = nx; (void)liship_memset(d->sx, 0, nx * sizeof(double), lisg.stream); }
	}
	if (!err) { int rc__ = liship_memcpy_h2d(d->sx, x, sizeof(double) * (size_t)A->n, lisg.stream); if (rc__) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc__); }
	if (!err) err = lisd_spmv_t(A, d->sx, d->sy);
	if (!err) { int rc__ = liship_memcpy_d2h(y, d->sy, sizeof(double) * (size_t)(lisg.nprocs > 1 ? A->n : A->np), lisg.stream); if (!rc__) rc__ = liship_stream_synchronize(lisg.stream); if (rc__) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc__); }
	if (err) {
		fprintf(stderr, "liblis_amd: lis_matvech_<fmt>(A, x[], y[]) failed (code %d) and has no error channel -- aborting\n", (int)err);
		abort();
	}
}

void lis_matvech_csr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_CSR, x, y); }
void lis_matvech_csc(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_CSC, x, y); }
void lis_matvech_ell(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_ELL, x, y); }
void lis_matvech_dia(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_DIA, x, y); }
void lis_matvech_jad(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_JAD, x, y); }
void lis_matvech_bsr(LIS_MATRIX A, LIS_SCALAR x[], LIS_SCALAR y[]) { raw_matvech(A, LIS_MATRIX_BSR, x, y); }
