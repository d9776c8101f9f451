/*
 * lis_scale.c -- diagonal scaling of the system before the Krylov loop (`-scale jacobi|symm_diag`).
 *
 * Reference: lis_matrix_scale, src/matrix/lis_matrix_ops.c:579-747, and the per-format loops
 * lis_matrix_scale_<fmt> / lis_matrix_scale_symm_<fmt> (lis_matrix_csr.c:640-700, _csc.c:375-430, _ell.c:735-830,
 * _dia.c:590-715, _jad.c:650-745, _bsr.c:840-955).  A one-off pass over the matrix: it runs on the host arrays
 * (the truth in both residency modes) and drops the HBM copy, which is rebuilt on the next product.
 * The rounding of every entry is the reference's, including its per-format association:
 *     jacobi      v *= d[row]                                   (all formats)
 *     symm_diag   CSR: (v*d[row])*d[col]   CSC: (v*d[col])*d[row]   ELL/DIA/JAD/BSR: v*(d[row]*d[col])
 * with d = 1/diag (jacobi) or 1/sqrt|diag| (symm_diag); b is scaled by d; A and b stay scaled afterwards
 * (A->is_scaled, B->is_scaled), exactly as the reference leaves them.
 */
#include "lis_internal.h"

static LIS_INT scale_values(LIS_MATRIX A, const double *d, int symm)
{
	const LIS_INT n = A->n;
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		for (LIS_INT i = 0; i < n; i++)
			for (LIS_INT j = A->ptr[i]; j < A->ptr[i + 1]; j++) {
				if (symm) A->value[j] = A->value[j] * d[i] * d[A->index[j]];
				else A->value[j] *= d[i];
			}
		break;
	case LIS_MATRIX_CSC:
		for (LIS_INT i = 0; i < A->np; i++)
			for (LIS_INT j = A->ptr[i]; j < A->ptr[i + 1]; j++) {
				if (symm) A->value[j] = A->value[j] * d[i] * d[A->index[j]];
				else A->value[j] *= d[A->index[j]];
			}
		break;
	case LIS_MATRIX_ELL:
		for (LIS_INT j = 0; j < A->maxnzr; j++)
			for (LIS_INT i = 0; i < n; i++) {
				const size_t k = (size_t)j * n + i;
				if (symm) A->value[k] *= d[i] * d[A->index[k]];
				else A->value[k] *= d[i];
			}
		break;
	case LIS_MATRIX_DIA:
		for (LIS_INT j = 0; j < A->nnd; j++) {
			const LIS_INT jj = A->index[j];
			const LIS_INT js = jj < 0 ? -jj : 0;
			const LIS_INT je = (lisg.nprocs > 1) ? (jj <= A->np - n ? n : (A->np - jj < n ? A->np - jj : n)) : (n - jj < n ? n - jj : n);
			for (LIS_INT i = js; i < je; i++) {
				const size_t k = (size_t)j * n + i;
				if (symm) A->value[k] *= d[i] * d[i + jj];
				else A->value[k] *= d[i];
			}
		}
		break;
	case LIS_MATRIX_JAD:
		for (LIS_INT j = 0; j < A->maxnzr; j++) {
			LIS_INT k = 0;
			for (LIS_INT i = A->ptr[j]; i < A->ptr[j + 1]; i++, k++) {
				if (symm) A->value[i] *= d[A->row[k]] * d[A->index[i]];
				else A->value[i] *= d[A->row[k]];
			}
		}
		break;
	case LIS_MATRIX_BSR: {
		const LIS_INT bnr = A->bnr, bnc = A->bnc, bs = bnr * bnc;
		for (LIS_INT bi = 0; bi < A->nr; bi++)
			for (LIS_INT bj = A->bptr[bi]; bj < A->bptr[bi + 1]; bj++) {
				const LIS_INT bjj = A->bindex[bj];
				for (LIS_INT j = 0; j < bnc; j++)
					for (LIS_INT i = 0; i < bnr; i++) {
						const size_t k = (size_t)bj * bs + (size_t)j * bnr + i;
						if (symm) A->value[k] *= d[bi * bnr + i] * d[bjj * bnc + j];
						else A->value[k] *= d[bi * bnr + i];
					}
			}
		break; }
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_scale(LIS_MATRIX A, LIS_VECTOR B, LIS_VECTOR D, LIS_INT action)
{
	if (!lisi_is_registered(A) || A->status < LIS_MATRIX_CSR) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is not assembled\n");
	if (MDEV(A)->device_only) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "scaling a matrix that lives in HBM only is not implemented\n");
	if (A->is_splited) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "split (D/L/U) matrices are not served\n");
	if (action != LIS_SCALE_JACOBI && action != LIS_SCALE_SYMM_DIAG) return LIS_SUCCESS;   /* the reference falls through too */
	const LIS_INT n = A->n, np = A->np;
	LISCHK(lis_matrix_get_diagonal(A, D));
	size_t need = (size_t)np + (size_t)A->pad;
	if (A->matrix_type == LIS_MATRIX_BSR) {
		const size_t a = (size_t)A->nr * A->bnr, b = (size_t)A->nc * A->bnc;
		if (a > need) need = a;
		if (b > need) need = b;
	}
	LISCHK(lisd_vec_to_host(D));
	if (VDEV(D)->hlen < need) {                       /* D->value grows like the reference's lis_realloc (:598-606), new entries zero */
		double *nv = (double *)calloc(need, sizeof(double));
		if (!nv) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)need);
		memcpy(nv, D->value, sizeof(double) * VDEV(D)->hlen);
		free(D->value);
		D->value = nv; VDEV(D)->hlen = need;
		if (D->np < np) D->np = np;
	}
	LISCHK(lisd_vec_to_host(B));
	double *d = D->value, *b = B->value;
	if (action == LIS_SCALE_SYMM_DIAG) {
		if (lisg.nprocs > 1 && A->commtable) {         /* ghosts of the diagonal, ref :596-608 */
			double *dd;
			lis_amd_vector_host_modified(D);
			LISCHK(lisd_vec_in(D, &dd));
			LISCHK(lisc_halo_device(A, dd));
			lis_amd_vector_device_modified(D);
			LISCHK(lisd_vec_to_host(D));
			d = D->value;
		}
		for (LIS_INT i = 0; i < np; i++) d[i] = 1.0 / sqrt(fabs(d[i]));
	} else {
		for (LIS_INT i = 0; i < n; i++) d[i] = 1.0 / d[i];
	}
	LISCHK(scale_values(A, d, action == LIS_SCALE_SYMM_DIAG));
	for (LIS_INT i = 0; i < n; i++) b[i] = b[i] * d[i];
	lis_amd_vector_host_modified(D);
	lis_amd_vector_host_modified(B);
	lisd_mat_free(A);                                  /* HBM copy (and its transpose) are stale */
	A->is_scaled = LIS_TRUE;
	B->is_scaled = LIS_TRUE;
	return LIS_SUCCESS;
}
