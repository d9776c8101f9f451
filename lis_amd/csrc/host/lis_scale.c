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

/* the split forms (D/L/U) the reference scales part by part: CSR (lis_matrix_csr.c:617-632, :661-676: D becomes 1) and BSR
 * (lis_matrix_bsr.c:820-855, :895-935: the diagonal blocks are scaled like the others -- by d[row]*d[ROW] in the symmetric case, as there) */
static LIS_INT scale_values_split(LIS_MATRIX A, const double *d, int symm)
{
	const LIS_INT n = A->n;
	if (A->matrix_type == LIS_MATRIX_CSR) {
		for (LIS_INT i = 0; i < n; i++) {
			A->D->value[i] = 1.0;
			for (int part = 0; part < 2; part++) {
				const LIS_MATRIX_CORE P = part ? A->U : A->L;
				for (LIS_INT j = P->ptr[i]; j < P->ptr[i + 1]; j++) {
					if (symm) P->value[j] = P->value[j] * d[i] * d[P->index[j]];
					else P->value[j] *= d[i];
				}
			}
		}
		return LIS_SUCCESS;
	}
	if (A->matrix_type == LIS_MATRIX_BSR) {
		const LIS_INT bnr = A->bnr, bnc = A->bnc, bs = bnr * bnc;
		for (LIS_INT bi = 0; bi < A->nr; bi++) {
			for (int part = 0; part < 2; part++) {
				const LIS_MATRIX_CORE P = part ? A->U : A->L;
				for (LIS_INT bj = P->bptr[bi]; bj < P->bptr[bi + 1]; bj++) {
					const LIS_INT bjj = P->bindex[bj];
					for (LIS_INT j = 0; j < bnc; j++)
						for (LIS_INT i = 0; i < bnr; i++) {
							const size_t k = (size_t)bj * bs + (size_t)j * bnr + i;
							if (symm) P->value[k] *= d[bi * bnr + i] * d[bjj * bnc + j];
							else P->value[k] *= d[bi * bnr + i];
						}
				}
			}
			for (LIS_INT j = 0; j < bnc; j++)
				for (LIS_INT i = 0; i < bnr; i++) {
					const size_t k = (size_t)bi * bs + (size_t)j * bnr + i;
					if (symm) A->D->value[k] *= d[bi * bnr + i] * d[bi * bnr + i];
					else A->D->value[k] *= d[bi * bnr + i];
				}
		}
		return LIS_SUCCESS;
	}
	return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "scaling a split (D/L/U) matrix is served for CSR and BSR storage\n");
}

static LIS_INT scale_values(LIS_MATRIX A, const double *d, int symm)
{
	const LIS_INT n = A->n;
	if (A->is_splited) return scale_values_split(A, d, symm);
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

/* A matrix whose arrays live in HBM only (lis_amd_matrix_set_csr_device, lis_amd_matrix_poisson3d): the same passes by kernels -- diagonal,
 * d = 1/diag or 1/sqrt|diag| (with the neighbours' entries behind it for the symmetric form: lis_matrix_ops.c:596-608), the rows' values,
 * b -- and the plan is rebuilt on the scaled values (value records of the unscaled matrix are gone). */
static LIS_INT scale_device_only(LIS_MATRIX A, LIS_VECTOR B, LIS_VECTOR D, LIS_INT action)
{
	lisd_mat *m = MDEV(A);
	if (A->matrix_type != LIS_MATRIX_CSR || m->type != LIS_MATRIX_CSR || A->is_splited)
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "matrices that live in HBM only are scaled in CSR storage\n");
	const LIS_INT n = A->n, np = A->np;
	const int symm = action == LIS_SCALE_SYMM_DIAG;
	double *dd, *db;
	if (D->np < np) D->np = np;
	LISCHK(lisd_vec_reserve(D, (size_t)np + (size_t)A->pad));
	LISCHK(lis_matrix_get_diagonal(A, D));             /* on the device for such a matrix */
	LISCHK(lisd_vec_in(D, &dd));
	if (symm) {
		if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dd));
		HIPCHK(liship_rsqrt_abs_f64(np, dd, lisg.stream));
	} else HIPCHK(liship_reciprocal_f64(n, dd, lisg.stream));
	LISCHK(lisd_vec_done(D));
	LISCHK(lisd_vec_in(D, &dd));
	HIPCHK(liship_csr_scale_f64(n, m->ptr, m->index, m->value, dd, symm, lisg.stream));
	LISCHK(lisd_vec_in(B, &db));
	HIPCHK(liship_pmul_f64(n, db, dd, db, lisg.stream));      /* b[i] = b[i]*d[i] */
	LISCHK(lisd_vec_done(B));
	/* everything derived from the old values goes: the plan (codes, patterns, value records), the transposed copy */
	if (m->plan) { (void)liship_csr_plan_destroy(m->plan); m->plan = NULL; }
	if (m->t_plan) { (void)liship_csr_plan_destroy(m->t_plan); m->t_plan = NULL; }
	(void)liship_free(m->t_ptr); (void)liship_free(m->t_index); (void)liship_free(m->t_value); (void)liship_free(m->t_diag);
	m->t_ptr = NULL; m->t_index = NULL; m->t_value = NULL; m->t_diag = NULL; m->t_ready = 0;
	/* ... and the renumbered transpose cache beside it (built from the UNSCALED values by a solve in a reordered plan's numbering: swap_transposed must never bring it back) */
	if (m->rt_plan) { (void)liship_csr_plan_destroy(m->rt_plan); m->rt_plan = NULL; }
	(void)liship_free(m->rt_ptr); (void)liship_free(m->rt_index); (void)liship_free(m->rt_value);
	m->rt_ptr = NULL; m->rt_index = NULL; m->rt_value = NULL; m->rt_ready = 0; m->rt_nnz = 0;
	m->reorder_tried = 0; m->served = 0;
	LISCHK(lisd_csr_plan_plain(&m->plan, n, m->ptr, m->index, m->value));
	A->is_scaled = LIS_TRUE;
	B->is_scaled = LIS_TRUE;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_scale(LIS_MATRIX A, LIS_VECTOR B, LIS_VECTOR D, LIS_INT action)
{
	if (!lisi_is_registered(A) || A->status < LIS_MATRIX_CSR) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is not assembled\n");
	if (A->is_splited && A->matrix_type != LIS_MATRIX_CSR && A->matrix_type != LIS_MATRIX_BSR)
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "scaling a split (D/L/U) matrix is served for CSR and BSR storage\n");
	if (action != LIS_SCALE_JACOBI && action != LIS_SCALE_SYMM_DIAG) return LIS_SUCCESS;   /* the reference falls through too */
	if (MDEV(A)->device_only) return scale_device_only(A, B, D, action);
	LISCHK(lisp_fill_matrix(A));
	const LIS_INT n = A->n, np = A->np;
	LISCHK(lis_matrix_get_diagonal(A, D));
	size_t need = (size_t)np + (size_t)A->pad;
	if (A->matrix_type == LIS_MATRIX_BSR) {
		const size_t a = (size_t)A->nr * A->bnr, b = (size_t)A->nc * A->bnc;
		if (a > need) need = a;
		if (b > need) need = b;
	}
	LISCHK(lisd_vec_host_write(D, 1));
	if (VDEV(D)->hlen < need) {                       /* D->value grows like the reference's lis_realloc (:598-606), new entries zero */
		LISCHK(lisp_grow(D, need));
		if (D->np < np) D->np = np;
	}
	LISCHK(lisd_vec_host_write(B, 1));
	double *d = D->value, *b = B->value;
	if (action == LIS_SCALE_SYMM_DIAG) {
		if (lisg.nprocs > 1 && A->commtable) {         /* ghosts of the diagonal, ref :596-608 */
			double *dd;
			lis_amd_vector_host_modified(D);
			LISCHK(lisd_vec_in(D, &dd));
			LISCHK(lisc_halo_device(A, dd));
			lis_amd_vector_device_modified(D);
			LISCHK(lisd_vec_host_write(D, 1));
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

/* ------------------------------------------------------------------ block-diagonal scaling of BSR storage
 * `-scale jacobi -storage bsr` (ref lis_solver.c:659-690): A is split, WD = D^-1 block by block (lis_matrix_diag_inverse,
 * lis_matrix_diag.c:750-806: Gaussian elimination without pivoting, lis_array_ge, src/array/lis_array.c:907-956; the rows of
 * the last block beyond n get a unit diagonal first), then A <- WD A (lis_matrix_bscale_bsr, lis_matrix_bsr.c:959-1066: D
 * becomes the identity, every L / U block is multiplied from the left) and b <- WD b (lis_matrix_diag_matvec, :810-895).
 * Every sum runs left to right over the block's columns, as the reference's expressions do.
 * The reference's bscale has cases for 1 x 1, 2 x 2 and 3 x 3 blocks only, and its 3 x 3 case copies 8 of the 9 entries back
 * (lis_matrix_bsr.c:1046,1058): for blocks of 3 and more it solves a system that is not the scaled one.  Here every block size
 * is scaled completely; the results carry the reference's bits for 1 x 1 and 2 x 2 (the default block size). */
static void block_inverse(LIS_INT bn, LIS_SCALAR *a, LIS_SCALAR *lu)
{
	const LIS_INT n = bn;
	memcpy(lu, a, sizeof(LIS_SCALAR) * (size_t)n * n);
	for (LIS_INT k = 0; k < n; k++) {                  /* LU without pivoting, the reciprocal of the pivot stored on the diagonal */
		lu[k + k * n] = 1.0 / lu[k + k * n];
		for (LIS_INT i = k + 1; i < n; i++) {
			const LIS_SCALAR t = lu[i + k * n] * lu[k + k * n];
			for (LIS_INT j = k + 1; j < n; j++) lu[i + j * n] -= t * lu[k + j * n];
			lu[i + k * n] = t;
		}
	}
	for (LIS_INT k = 0; k < n; k++) {                  /* column k of the inverse: forward, then backward substitution */
		for (LIS_INT i = 0; i < n; i++) {
			LIS_SCALAR t = (i == k);
			for (LIS_INT j = 0; j < i; j++) t -= lu[i + j * n] * a[j + k * n];
			a[i + k * n] = t;
		}
		for (LIS_INT i = n - 1; i >= 0; i--) {
			LIS_SCALAR t = a[i + k * n];
			for (LIS_INT j = i + 1; j < n; j++) t -= lu[i + j * n] * a[j + k * n];
			a[k * n + i] = t * lu[i + i * n];
		}
	}
}

static void block_left_multiply(LIS_INT bn, const LIS_SCALAR *d, LIS_SCALAR *v, LIS_SCALAR *tmp)
{	/* v <- d v, column-major blocks: out(i,j) = d(i,0) v(0,j) + d(i,1) v(1,j) + ... left to right */
	for (LIS_INT j = 0; j < bn; j++)
		for (LIS_INT i = 0; i < bn; i++) {
			LIS_SCALAR t = d[i] * v[j * bn];
			for (LIS_INT k = 1; k < bn; k++) t += d[i + k * bn] * v[k + j * bn];
			tmp[i + j * bn] = t;
		}
	memcpy(v, tmp, sizeof(LIS_SCALAR) * (size_t)bn * bn);
}

LIS_INT lisi_matrix_bscale_bsr(LIS_MATRIX A, LIS_VECTOR B)
{
	if (A->matrix_type != LIS_MATRIX_BSR || A->bnr != A->bnc) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "block scaling needs square BSR blocks\n");
	LISCHK(lis_matrix_split(A));
	const LIS_INT nr = A->nr, bn = A->bnr, n = A->n;
	const size_t bs = (size_t)bn * bn;
	LIS_SCALAR *wd = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * bs * (size_t)(nr > 0 ? nr : 1));
	LIS_SCALAR *lu = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * bs * 2);
	if (!wd || !lu) { free(wd); free(lu); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)(bs * nr)); }
	memcpy(wd, A->D->value, sizeof(LIS_SCALAR) * bs * (size_t)nr);
	if (bn == 1) for (LIS_INT i = 0; i < nr; i++) wd[i] = 1.0 / wd[i];
	else {
		const LIS_INT k = n % bn;
		if (k != 0) for (LIS_INT i = bn - 1; i >= k; i--) wd[bs * (size_t)(nr - 1) + (size_t)i * (bn + 1)] = 1.0;
		for (LIS_INT i = 0; i < nr; i++) block_inverse(bn, wd + bs * (size_t)i, lu);
	}
	LIS_SCALAR *tmp = lu + bs;
	for (LIS_INT bi = 0; bi < nr; bi++) {
		const LIS_SCALAR *d = wd + bs * (size_t)bi;
		LIS_SCALAR *dd = A->D->value + bs * (size_t)bi;
		for (LIS_INT j = 0; j < bn; j++) for (LIS_INT i = 0; i < bn; i++) dd[i + j * bn] = (i == j) ? 1.0 : 0.0;
		if (bn == 1) {
			for (LIS_INT bj = A->L->bptr[bi]; bj < A->L->bptr[bi + 1]; bj++) A->L->value[bj] *= d[0];
			for (LIS_INT bj = A->U->bptr[bi]; bj < A->U->bptr[bi + 1]; bj++) A->U->value[bj] *= d[0];
		} else {
			for (LIS_INT bj = A->L->bptr[bi]; bj < A->L->bptr[bi + 1]; bj++) block_left_multiply(bn, d, A->L->value + bs * (size_t)bj, tmp);
			for (LIS_INT bj = A->U->bptr[bi]; bj < A->U->bptr[bi + 1]; bj++) block_left_multiply(bn, d, A->U->value + bs * (size_t)bj, tmp);
		}
	}
	/* b <- WD b; the rows of the last block beyond n read the vector's padding (zeros) */
	{ const LIS_INT werr = lisd_vec_host_write(B, 1); if (werr) { free(wd); free(lu); return werr; } }
	{
		const size_t have = VDEV(B)->hlen;
		LIS_SCALAR *t = (LIS_SCALAR *)calloc((size_t)nr * bn + 1, sizeof(LIS_SCALAR));
		if (!t) { free(wd); free(lu); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", nr * bn); }
		for (LIS_INT bi = 0; bi < nr; bi++) {
			const LIS_SCALAR *d = wd + bs * (size_t)bi;
			for (LIS_INT i = 0; i < bn; i++) {
				const size_t c0 = (size_t)bi * bn;
				LIS_SCALAR acc = (bn == 1) ? (c0 < have ? B->value[c0] : 0.0) * d[0] : d[i] * (c0 < have ? B->value[c0] : 0.0);
				for (LIS_INT k = 1; k < bn; k++) acc += d[i + k * bn] * (c0 + k < have ? B->value[c0 + k] : 0.0);
				t[c0 + i] = acc;
			}
		}
		for (LIS_INT i = 0; i < n; i++) B->value[i] = t[i];
		free(t);
	}
	free(wd); free(lu);
	lis_amd_vector_host_modified(B);
	lisd_mat_free(A);                    /* the HBM copy holds the unscaled parts */
	return LIS_SUCCESS;
}
