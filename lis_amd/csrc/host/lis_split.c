/*
 * lis_split.c -- the split form A = L + D + U of an assembled matrix.
 *
 * lis_matrix_split (ref src/matrix/lis_matrix_ops.c:860) leaves the strictly lower part in A->L, the strictly upper part in
 * A->U (LIS_MATRIX_CORE, the layout of A's own storage format) and the diagonal in A->D (LIS_MATRIX_DIAG), keeps A's own
 * arrays, and raises A->is_splited; from then on lis_matvec adds a row as  D x, then the L entries, then the U entries
 * (src/matvec/lis_matvec_<fmt>.c, the is_splited branches) -- another rounding sequence than the unsplit product.  In
 * scope it is reached by `-scale jacobi -storage bsr` (lis_solver.c:659-690, lis_matrix_bscale_bsr); the symbol is
 * exported for callers that link the internal header of the reference.
 *
 * Layouts follow the reference's per-format routines, restated (not copied):
 *   CSR  lis_matrix_csr.c:765   CSC  lis_matrix_csc.c:493   ELL  lis_matrix_ell.c:324   DIA  lis_matrix_dia.c:782
 *   JAD  lis_matrix_jad.c:815   BSR  lis_matrix_bsr.c:1131 (square blocks only, as there)
 * One deliberate difference: the reference mallocs D->value and writes only the rows that HAVE a diagonal entry; here the
 * array starts zeroed, so a missing diagonal is 0 instead of heap contents.
 *
 * The HBM copy of a split matrix is built by lis_device.c from lisi_split_rows(): every row as ONE chain of terms in the
 * order the reference adds them, which the CSR kernels sum from -0.0 ("the first product initialises the sum").
 */
#include "lis_internal.h"

/* ------------------------------------------------------------------ containers */
static LIS_MATRIX_DIAG diag_new(LIS_MATRIX A)
{	/* ref lis_matrix_diag.c:338-448 (lis_matrix_diag_duplicateM) */
	LIS_MATRIX_DIAG D = (LIS_MATRIX_DIAG)calloc(1, sizeof(struct LIS_MATRIX_DIAG_STRUCT));
	if (!D) return NULL;
	size_t count = (size_t)(A->np > 0 ? A->np : 1);
	D->nr = A->n;
	if (A->matrix_type == LIS_MATRIX_BSR) {
		count = (size_t)A->nr * (size_t)A->bnr * (size_t)A->bnc;
		if (count == 0) count = 1;
		D->bn = A->bnr;
		D->nr = A->nr;
	}
	D->value = (LIS_SCALAR *)calloc(count, sizeof(LIS_SCALAR));
	if (!D->value) { free(D); return NULL; }
	D->status = LIS_MATRIX_NULL;        /* lis_matrix_diag_init, ref :53-64 */
	D->is_destroy = LIS_TRUE;
	D->n = A->n; D->gn = A->gn; D->np = A->np;
	D->comm = A->comm; D->my_rank = A->my_rank; D->nprocs = A->nprocs;
	D->is = A->is; D->ie = A->ie; D->origin = A->origin;
	return D;
}

static void diag_free(LIS_MATRIX_DIAG D)
{
	if (!D) return;
	free(D->value); free(D->work); free(D->bns); free(D->ptr); free(D->ranges);
	free(D);
}

static void core_free(LIS_MATRIX_CORE c)
{	/* ref lis_matrix.c:335-356 */
	if (!c) return;
	free(c->ptr); free(c->row); free(c->col); free(c->index); free(c->bptr); free(c->bindex); free(c->value); free(c->work);
	free(c);
}

void lisi_matrix_dlu_destroy(LIS_MATRIX A)
{	/* ref lis_matrix.c:360-376 */
	if (!A) return;
	diag_free(A->D); core_free(A->L); core_free(A->U);
	A->D = NULL; A->L = NULL; A->U = NULL;
	A->is_splited = LIS_FALSE;
}

static void *zalloc(size_t count, size_t size) { return calloc(count ? count : 1, size); }

/* ------------------------------------------------------------------ per-format splits */
/* CSR rows / CSC columns: entries left of the diagonal position go to L, right of it to U, the (last) one on it to D */
static LIS_INT split_compressed(LIS_MATRIX A, LIS_MATRIX_CORE L, LIS_MATRIX_CORE U, LIS_MATRIX_DIAG D, LIS_INT lines)
{
	LIS_INT nl = 0, nu = 0;
	for (LIS_INT i = 0; i < lines; i++)
		for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) { if (A->index[k] < i) nl++; else if (A->index[k] > i) nu++; }
	L->ptr = (LIS_INT *)zalloc((size_t)lines + 1, sizeof(LIS_INT)); U->ptr = (LIS_INT *)zalloc((size_t)lines + 1, sizeof(LIS_INT));
	L->index = (LIS_INT *)zalloc((size_t)nl, sizeof(LIS_INT));       U->index = (LIS_INT *)zalloc((size_t)nu, sizeof(LIS_INT));
	L->value = (LIS_SCALAR *)zalloc((size_t)nl, sizeof(LIS_SCALAR)); U->value = (LIS_SCALAR *)zalloc((size_t)nu, sizeof(LIS_SCALAR));
	if (!L->ptr || !U->ptr || !L->index || !U->index || !L->value || !U->value) return LIS_ERR_OUT_OF_MEMORY;
	nl = nu = 0;
	for (LIS_INT i = 0; i < lines; i++) {
		for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) {
			const LIS_INT c = A->index[k];
			if (c < i)      { L->index[nl] = c; L->value[nl] = A->value[k]; nl++; }
			else if (c > i) { U->index[nu] = c; U->value[nu] = A->value[k]; nu++; }
			else D->value[i] = A->value[k];
		}
		L->ptr[i + 1] = nl; U->ptr[i + 1] = nu;
	}
	L->nnz = nl; U->nnz = nu;
	return LIS_SUCCESS;
}

static LIS_INT split_ell(LIS_MATRIX A, LIS_MATRIX_CORE L, LIS_MATRIX_CORE U, LIS_MATRIX_DIAG D)
{
	const LIS_INT n = A->n, mx = A->maxnzr;
	LIS_INT lmax = 0, umax = 0;
	for (LIS_INT i = 0; i < n; i++) {
		LIS_INT lc = 0, uc = 0;
		for (LIS_INT j = 0; j < mx; j++) { const LIS_INT c = A->index[(size_t)j * n + i]; if (c < i) lc++; else if (c > i) uc++; }
		if (lc > lmax) lmax = lc;
		if (uc > umax) umax = uc;
	}
	L->index = (LIS_INT *)zalloc((size_t)lmax * n, sizeof(LIS_INT)); L->value = (LIS_SCALAR *)zalloc((size_t)lmax * n, sizeof(LIS_SCALAR));
	U->index = (LIS_INT *)zalloc((size_t)umax * n, sizeof(LIS_INT)); U->value = (LIS_SCALAR *)zalloc((size_t)umax * n, sizeof(LIS_SCALAR));
	if (!L->index || !L->value || !U->index || !U->value) return LIS_ERR_OUT_OF_MEMORY;
	for (LIS_INT j = 0; j < lmax; j++) for (LIS_INT i = 0; i < n; i++) L->index[(size_t)j * n + i] = i;      /* padding: value 0, column = row */
	for (LIS_INT j = 0; j < umax; j++) for (LIS_INT i = 0; i < n; i++) U->index[(size_t)j * n + i] = i;
	for (LIS_INT i = 0; i < n; i++) {
		LIS_INT lc = 0, uc = 0;
		for (LIS_INT j = 0; j < mx; j++) {
			const size_t at = (size_t)j * n + i;
			const LIS_INT c = A->index[at];
			if (c < i)      { L->index[(size_t)lc * n + i] = c; L->value[(size_t)lc * n + i] = A->value[at]; lc++; }
			else if (c > i) { U->index[(size_t)uc * n + i] = c; U->value[(size_t)uc * n + i] = A->value[at]; uc++; }
			else if (A->value[at] != 0.0) D->value[i] = A->value[at];      /* the padding of A sits on the diagonal with value 0 */
		}
	}
	L->maxnzr = lmax; U->maxnzr = umax;
	return LIS_SUCCESS;
}

static LIS_INT split_dia(LIS_MATRIX A, LIS_MATRIX_CORE L, LIS_MATRIX_CORE U, LIS_MATRIX_DIAG D)
{	/* one-chunk layout value[d*n + i] */
	const LIS_INT n = A->n;
	LIS_INT nl = 0, nu = 0;
	for (LIS_INT d = 0; d < A->nnd; d++) { if (A->index[d] < 0) nl++; else if (A->index[d] > 0) nu++; }
	L->index = (LIS_INT *)zalloc((size_t)nl, sizeof(LIS_INT)); L->value = (LIS_SCALAR *)zalloc((size_t)nl * n, sizeof(LIS_SCALAR));
	U->index = (LIS_INT *)zalloc((size_t)nu, sizeof(LIS_INT)); U->value = (LIS_SCALAR *)zalloc((size_t)nu * n, sizeof(LIS_SCALAR));
	if (!L->index || !L->value || !U->index || !U->value) return LIS_ERR_OUT_OF_MEMORY;
	nl = nu = 0;
	for (LIS_INT d = 0; d < A->nnd; d++) {
		const LIS_SCALAR *src = A->value + (size_t)d * n;
		if (A->index[d] < 0)      { L->index[nl] = A->index[d]; memcpy(L->value + (size_t)nl * n, src, sizeof(LIS_SCALAR) * (size_t)n); nl++; }
		else if (A->index[d] > 0) { U->index[nu] = A->index[d]; memcpy(U->value + (size_t)nu * n, src, sizeof(LIS_SCALAR) * (size_t)n); nu++; }
		else memcpy(D->value, src, sizeof(LIS_SCALAR) * (size_t)n);
	}
	L->nnd = nl; U->nnd = nu;
	return LIS_SUCCESS;
}

static LIS_INT split_bsr(LIS_MATRIX A, LIS_MATRIX_CORE L, LIS_MATRIX_CORE U, LIS_MATRIX_DIAG D)
{
	const LIS_INT nr = A->nr;
	const size_t bs = (size_t)A->bnr * (size_t)A->bnc;
	LIS_INT nl = 0, nu = 0;
	for (LIS_INT i = 0; i < nr; i++)
		for (LIS_INT k = A->bptr[i]; k < A->bptr[i + 1]; k++) { if (A->bindex[k] < i) nl++; else if (A->bindex[k] > i) nu++; }
	L->bptr = (LIS_INT *)zalloc((size_t)nr + 1, sizeof(LIS_INT)); U->bptr = (LIS_INT *)zalloc((size_t)nr + 1, sizeof(LIS_INT));
	L->bindex = (LIS_INT *)zalloc((size_t)nl, sizeof(LIS_INT));   U->bindex = (LIS_INT *)zalloc((size_t)nu, sizeof(LIS_INT));
	L->value = (LIS_SCALAR *)zalloc((size_t)nl * bs, sizeof(LIS_SCALAR)); U->value = (LIS_SCALAR *)zalloc((size_t)nu * bs, sizeof(LIS_SCALAR));
	if (!L->bptr || !U->bptr || !L->bindex || !U->bindex || !L->value || !U->value) return LIS_ERR_OUT_OF_MEMORY;
	nl = nu = 0;
	for (LIS_INT i = 0; i < nr; i++) {
		for (LIS_INT k = A->bptr[i]; k < A->bptr[i + 1]; k++) {
			const LIS_SCALAR *src = A->value + bs * (size_t)k;
			if (A->bindex[k] < i)      { L->bindex[nl] = A->bindex[k]; memcpy(L->value + bs * (size_t)nl, src, sizeof(LIS_SCALAR) * bs); nl++; }
			else if (A->bindex[k] > i) { U->bindex[nu] = A->bindex[k]; memcpy(U->value + bs * (size_t)nu, src, sizeof(LIS_SCALAR) * bs); nu++; }
			else memcpy(D->value + bs * (size_t)i, src, sizeof(LIS_SCALAR) * bs);
		}
		L->bptr[i + 1] = nl; U->bptr[i + 1] = nu;
	}
	L->bnr = U->bnr = A->bnr; L->bnc = U->bnc = A->bnc; L->nr = U->nr = nr; L->nc = U->nc = A->nc;
	L->bnnz = nl; U->bnnz = nu;
	return LIS_SUCCESS;
}

/* JAD (one chunk): L and U are jagged matrices of their own.  Their row permutations come from sorting the rows, in the
 * order A's permutation lists them, by their L / U entry count with the reference's descending quicksort (lis_sortr_ii,
 * src/system/lis_sort.c -- unstable, so the tie order is part of the layout; restated in lisi_sortr_ii, lis_convert.c). */
static LIS_INT split_jad(LIS_MATRIX A, LIS_MATRIX_CORE L, LIS_MATRIX_CORE U, LIS_MATRIX_DIAG D)
{
	const LIS_INT n = A->n, mx = A->maxnzr;
	LIS_INT *lc = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT)), *uc = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT));
	LIS_INT *lpos = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT)), *upos = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT));
	LIS_INT err = LIS_ERR_OUT_OF_MEMORY, lnnz = 0, unnz = 0, lmax = 0, umax = 0;
	if (!lc || !uc || !lpos || !upos) goto out;
	/* counts per POSITION k of A's permuted order (slot k of every jagged diagonal belongs to row A->row[k]) */
	for (LIS_INT j = 0; j < mx; j++) {
		LIS_INT k = 0;
		for (LIS_INT i = A->ptr[j]; i < A->ptr[j + 1]; i++, k++) {
			if (A->index[i] < A->row[k])      { lnnz++; lc[k]++; }
			else if (A->index[i] > A->row[k]) { unnz++; uc[k]++; }
		}
	}
	for (LIS_INT i = 0; i < n; i++) { if (lc[i] > lmax) lmax = lc[i]; if (uc[i] > umax) umax = uc[i]; }
	L->row = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT)); U->row = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT));
	L->ptr = (LIS_INT *)zalloc((size_t)lmax + 1, sizeof(LIS_INT)); U->ptr = (LIS_INT *)zalloc((size_t)umax + 1, sizeof(LIS_INT));
	L->index = (LIS_INT *)zalloc((size_t)lnnz, sizeof(LIS_INT)); U->index = (LIS_INT *)zalloc((size_t)unnz, sizeof(LIS_INT));
	L->value = (LIS_SCALAR *)zalloc((size_t)lnnz, sizeof(LIS_SCALAR)); U->value = (LIS_SCALAR *)zalloc((size_t)unnz, sizeof(LIS_SCALAR));
	if (!L->row || !U->row || !L->ptr || !U->ptr || !L->index || !U->index || !L->value || !U->value) goto out;
	for (LIS_INT i = 0; i < n; i++) {
		L->row[i] = A->row[i]; U->row[i] = A->row[i];
		for (LIS_INT j = 0; j < lc[i]; j++) L->ptr[j + 1]++;
		for (LIS_INT j = 0; j < uc[i]; j++) U->ptr[j + 1]++;
	}
	if (n > 0) { lisi_sortr_ii(0, n - 1, lc, L->row); lisi_sortr_ii(0, n - 1, uc, U->row); }
	for (LIS_INT j = 0; j < lmax; j++) L->ptr[j + 1] += L->ptr[j];
	for (LIS_INT j = 0; j < umax; j++) U->ptr[j + 1] += U->ptr[j];
	for (LIS_INT i = 0; i < n; i++) { lc[i] = 0; uc[i] = 0; lpos[L->row[i]] = i; upos[U->row[i]] = i; }     /* row -> slot in the new orders */
	for (LIS_INT j = 0; j < mx; j++) {
		LIS_INT k = 0;
		for (LIS_INT i = A->ptr[j]; i < A->ptr[j + 1]; i++, k++) {
			const LIS_INT r = A->row[k];
			if (A->index[i] < r)      { const LIS_INT at = L->ptr[lc[r]] + lpos[r]; lc[r]++; L->index[at] = A->index[i]; L->value[at] = A->value[i]; }
			else if (A->index[i] > r) { const LIS_INT at = U->ptr[uc[r]] + upos[r]; uc[r]++; U->index[at] = A->index[i]; U->value[at] = A->value[i]; }
			else D->value[r] = A->value[i];
		}
	}
	L->nnz = lnnz; U->nnz = unnz; L->maxnzr = lmax; U->maxnzr = umax;
	err = LIS_SUCCESS;
out:
	free(lc); free(uc); free(lpos); free(upos);
	return err;
}

LIS_INT lis_matrix_split(LIS_MATRIX A)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_ASSEMBLED));
	if (A->is_splited) return LIS_SUCCESS;
	LISCHK(lisp_fill_matrix(A));          /* the parts are built from the host arrays */
	if (MDEV(A)->device_only) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "matrix lives in HBM only: split the host matrix before uploading\n");
	if (A->matrix_type == LIS_MATRIX_BSR && A->bnr != A->bnc) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "split of non-square blocks is not implemented\n");   /* ref lis_matrix_bsr.c:1164 */
	LIS_MATRIX_CORE L = (LIS_MATRIX_CORE)calloc(1, sizeof(struct LIS_MATRIX_CORE_STRUCT));
	LIS_MATRIX_CORE U = (LIS_MATRIX_CORE)calloc(1, sizeof(struct LIS_MATRIX_CORE_STRUCT));
	LIS_MATRIX_DIAG D = diag_new(A);
	LIS_INT err = (L && U && D) ? LIS_SUCCESS : LIS_ERR_OUT_OF_MEMORY;
	if (!err) switch (A->matrix_type) {
	case LIS_MATRIX_CSR: err = split_compressed(A, L, U, D, A->n); break;
	case LIS_MATRIX_CSC: err = split_compressed(A, L, U, D, A->np); break;
	case LIS_MATRIX_ELL: err = split_ell(A, L, U, D); break;
	case LIS_MATRIX_DIA: err = split_dia(A, L, U, D); break;
	case LIS_MATRIX_JAD: err = split_jad(A, L, U, D); break;
	case LIS_MATRIX_BSR: err = split_bsr(A, L, U, D); break;
	default: err = LIS_ERR_NOT_IMPLEMENTED; break;
	}
	if (err) {
		core_free(L); core_free(U); diag_free(D);
		if (err == LIS_ERR_OUT_OF_MEMORY) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", A->nnz);
		return LISI_ERR(err, "storage format %D cannot be split\n", A->matrix_type);
	}
	A->L = L; A->U = U; A->D = D;
	A->is_splited = LIS_TRUE;
	lisd_mat_free(A);                   /* the HBM copy adds rows in the unsplit order: rebuilt on the next product */
	return LIS_SUCCESS;
}

/* ref lis_matrix_ops.c:1052-1112.  CSR and BSR: A's arrays are REBUILT from the parts as lis_matrix_merge_csr / _bsr do
 * (lis_matrix_csr.c merge, lis_matrix_bsr.c:1337-1395) -- every row lists its L entries, then the diagonal (an entry the row may
 * not have had), then its U entries, with the parts' CURRENT values: after -scale jacobi -storage bsr the parts hold the scaled
 * system (lis_scale.c), and a later lis_matrix_convert / lis_matrix_copy / second lis_solve must see that system, not the arrays
 * the split started from.  The other formats' parts are never modified by anything served here, and their own arrays were
 * never dropped (nor in the reference: the destroy in lis_matrix_split is commented out), so merging them only retires the parts. */
static LIS_INT merge_csr(LIS_MATRIX A)
{
	const LIS_INT n = A->n;
	const LIS_INT nnz = A->L->nnz + A->U->nnz + n;
	LIS_INT *ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * ((size_t)n + 1));
	LIS_INT *index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nnz > 0 ? nnz : 1));
	LIS_SCALAR *value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(nnz > 0 ? nnz : 1));
	if (!ptr || !index || !value) { free(ptr); free(index); free(value); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", nnz); }
	LIS_INT at = 0;
	ptr[0] = 0;
	for (LIS_INT i = 0; i < n; i++) {
		for (LIS_INT j = A->L->ptr[i]; j < A->L->ptr[i + 1]; j++) { index[at] = A->L->index[j]; value[at] = A->L->value[j]; at++; }
		index[at] = i; value[at] = A->D->value[i]; at++;
		for (LIS_INT j = A->U->ptr[i]; j < A->U->ptr[i + 1]; j++) { index[at] = A->U->index[j]; value[at] = A->U->value[j]; at++; }
		ptr[i + 1] = at;
	}
	if (A->is_destroy) { if (!lisp_free_array(A->ptr)) free(A->ptr); if (!lisp_free_array(A->index)) free(A->index); if (!lisp_free_array(A->value)) free(A->value); }
	A->ptr = ptr; A->index = index; A->value = value; A->nnz = at;
	A->is_destroy = LIS_TRUE;            /* the rebuilt arrays are the library's, whoever owned the old ones */
	return LIS_SUCCESS;
}

static LIS_INT merge_bsr(LIS_MATRIX A)
{
	const LIS_INT nr = A->nr;
	const size_t bs = (size_t)A->bnr * A->bnc;
	const LIS_INT bnnz = A->L->bnnz + A->U->bnnz + nr;
	LIS_INT *bptr = (LIS_INT *)malloc(sizeof(LIS_INT) * ((size_t)nr + 1));
	LIS_INT *bindex = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(bnnz > 0 ? bnnz : 1));
	LIS_SCALAR *value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * bs * (size_t)(bnnz > 0 ? bnnz : 1));
	if (!bptr || !bindex || !value) { free(bptr); free(bindex); free(value); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", bnnz); }
	LIS_INT at = 0;
	bptr[0] = 0;
	for (LIS_INT i = 0; i < nr; i++) {
		for (LIS_INT j = A->L->bptr[i]; j < A->L->bptr[i + 1]; j++) { bindex[at] = A->L->bindex[j]; memcpy(value + bs * (size_t)at, A->L->value + bs * (size_t)j, sizeof(LIS_SCALAR) * bs); at++; }
		bindex[at] = i; memcpy(value + bs * (size_t)at, A->D->value + bs * (size_t)i, sizeof(LIS_SCALAR) * bs); at++;
		for (LIS_INT j = A->U->bptr[i]; j < A->U->bptr[i + 1]; j++) { bindex[at] = A->U->bindex[j]; memcpy(value + bs * (size_t)at, A->U->value + bs * (size_t)j, sizeof(LIS_SCALAR) * bs); at++; }
		bptr[i + 1] = at;
	}
	if (A->is_destroy) { if (!lisp_free_array(A->bptr)) free(A->bptr); if (!lisp_free_array(A->bindex)) free(A->bindex); if (!lisp_free_array(A->value)) free(A->value); }
	A->bptr = bptr; A->bindex = bindex; A->value = value; A->bnnz = at;
	A->is_destroy = LIS_TRUE;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_merge(LIS_MATRIX A)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_ASSEMBLED));
	if (!A->is_splited || A->is_save) return LIS_SUCCESS;
	if (A->matrix_type == LIS_MATRIX_CSR) LISCHK(merge_csr(A));
	else if (A->matrix_type == LIS_MATRIX_BSR && A->bnr == A->bnc) LISCHK(merge_bsr(A));
	lisi_matrix_dlu_destroy(A);
	lisd_mat_free(A);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ the split product as one chain per row
 * Terms of row r in the order the reference's is_splited branches add them:
 *   CSR  lis_matvec_csr.c:64-89     D, L row, U row
 *   CSC  lis_matvec_csc.c:65-90     D, then column by column (ascending) its L entry / U entry of that row
 *   ELL  lis_matvec_ell.c:56-87     D, L slots 0..lmaxnzr-1 (padding included), U slots
 *   DIA  lis_matvec_dia.c:56-123    D, L diagonals in stored order (where the column exists), U diagonals
 *   BSR  lis_matvec_bsr.c:293-325 (2x2; 3x3 :453, 4x4 :644, 1x1 :159): the block row of D column by column, then the L blocks,
 *        then the U blocks, each column by column; blocks larger than 4 take the generic routine (:70-118), which starts
 *        the sums at +0.0 instead of at the first product (*from_zero)
 * JAD is not a chain: lis_matvec_jad.c:60-140 forms (D x + sum of L) + sum of U with both partial sums started at 0 --
 * lis_device.c multiplies by L and by U separately and combines.
 * Returns CSR arrays over `*rows` rows (BSR: nr*bnr, padding rows included); the caller frees them. */
LIS_INT lisi_split_rows(LIS_MATRIX A, LIS_INT *rows, LIS_INT **optr, LIS_INT **oidx, LIS_SCALAR **oval, int *from_zero)
{
	const LIS_INT n = A->n;
	LIS_MATRIX_CORE L = A->L, U = A->U;
	const LIS_SCALAR *dv = A->D->value;
	LIS_INT nrows = n;
	size_t total = 0;
	*from_zero = 0;
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR: case LIS_MATRIX_CSC: total = (size_t)n + L->nnz + U->nnz; break;
	case LIS_MATRIX_ELL: total = (size_t)n * (1 + (size_t)L->maxnzr + U->maxnzr); break;
	case LIS_MATRIX_DIA: total = (size_t)n * (1 + (size_t)L->nnd + U->nnd); break;
	case LIS_MATRIX_BSR: nrows = A->nr * A->bnr; total = ((size_t)A->nr + L->bnnz + U->bnnz) * A->bnr * A->bnc; *from_zero = A->bnr > 4; break;
	default: return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "no chain form for storage format %D\n", A->matrix_type);
	}
	LIS_INT *ptr = (LIS_INT *)zalloc((size_t)nrows + 1, sizeof(LIS_INT));
	LIS_INT *idx = (LIS_INT *)zalloc(total, sizeof(LIS_INT));
	LIS_SCALAR *val = (LIS_SCALAR *)zalloc(total, sizeof(LIS_SCALAR));
	if (!ptr || !idx || !val) { free(ptr); free(idx); free(val); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)total); }
	LIS_INT k = 0;
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		for (LIS_INT r = 0; r < n; r++) {
			idx[k] = r; val[k++] = dv[r];
			for (LIS_INT j = L->ptr[r]; j < L->ptr[r + 1]; j++) { idx[k] = L->index[j]; val[k++] = L->value[j]; }
			for (LIS_INT j = U->ptr[r]; j < U->ptr[r + 1]; j++) { idx[k] = U->index[j]; val[k++] = U->value[j]; }
			ptr[r + 1] = k;
		}
		break;
	case LIS_MATRIX_CSC: {
		/* count per output row, then fill by walking the columns in the reference's order (rows >= n cannot occur: index < n) */
		for (LIS_INT r = 0; r < n; r++) ptr[r + 1] = 1;
		for (LIS_INT j = 0; j < L->nnz; j++) ptr[L->index[j] + 1]++;
		for (LIS_INT j = 0; j < U->nnz; j++) ptr[U->index[j] + 1]++;
		for (LIS_INT r = 0; r < n; r++) ptr[r + 1] += ptr[r];
		LIS_INT *fill = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT));
		if (!fill) { free(ptr); free(idx); free(val); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", n); }
		for (LIS_INT r = 0; r < n; r++) { fill[r] = ptr[r]; idx[fill[r]] = r; val[fill[r]++] = dv[r]; }
		for (LIS_INT c = 0; c < A->np; c++) {
			for (LIS_INT j = L->ptr[c]; j < L->ptr[c + 1]; j++) { const LIS_INT at = fill[L->index[j]]++; idx[at] = c; val[at] = L->value[j]; }
			for (LIS_INT j = U->ptr[c]; j < U->ptr[c + 1]; j++) { const LIS_INT at = fill[U->index[j]]++; idx[at] = c; val[at] = U->value[j]; }
		}
		free(fill);
		break;
	}
	case LIS_MATRIX_ELL:
		for (LIS_INT r = 0; r < n; r++) {
			idx[k] = r; val[k++] = dv[r];
			for (LIS_INT j = 0; j < L->maxnzr; j++) { idx[k] = L->index[(size_t)j * n + r]; val[k++] = L->value[(size_t)j * n + r]; }
			for (LIS_INT j = 0; j < U->maxnzr; j++) { idx[k] = U->index[(size_t)j * n + r]; val[k++] = U->value[(size_t)j * n + r]; }
			ptr[r + 1] = k;
		}
		break;
	case LIS_MATRIX_DIA:
		for (LIS_INT r = 0; r < n; r++) {
			idx[k] = r; val[k++] = dv[r];
			for (LIS_INT d = 0; d < L->nnd; d++) { const LIS_INT c = r + L->index[d]; if (c >= 0 && c < A->np) { idx[k] = c; val[k++] = L->value[(size_t)d * n + r]; } }
			for (LIS_INT d = 0; d < U->nnd; d++) { const LIS_INT c = r + U->index[d]; if (c >= 0 && c < A->np) { idx[k] = c; val[k++] = U->value[(size_t)d * n + r]; } }
			ptr[r + 1] = k;
		}
		break;
	case LIS_MATRIX_BSR: {
		const LIS_INT bnr = A->bnr, bnc = A->bnc;
		const size_t bs = (size_t)bnr * bnc;
		for (LIS_INT bi = 0; bi < A->nr; bi++)
			for (LIS_INT ii = 0; ii < bnr; ii++) {
				for (LIS_INT j = 0; j < bnc; j++) { idx[k] = bi * bnc + j; val[k++] = dv[bs * bi + (size_t)j * bnr + ii]; }
				for (LIS_INT bc = L->bptr[bi]; bc < L->bptr[bi + 1]; bc++)
					for (LIS_INT j = 0; j < bnc; j++) { idx[k] = L->bindex[bc] * bnc + j; val[k++] = L->value[bs * bc + (size_t)j * bnr + ii]; }
				for (LIS_INT bc = U->bptr[bi]; bc < U->bptr[bi + 1]; bc++)
					for (LIS_INT j = 0; j < bnc; j++) { idx[k] = U->bindex[bc] * bnc + j; val[k++] = U->value[bs * bc + (size_t)j * bnr + ii]; }
				ptr[bi * bnr + ii + 1] = k;
			}
		break;
	}
	default: break;
	}
	*rows = nrows; *optr = ptr; *oidx = idx; *oval = val;
	return LIS_SUCCESS;
}

/* one jagged part (L or U of a split JAD matrix) as CSR rows in the ORIGINAL row order, entries in jagged-diagonal order:
 * the order lis_matvec_jad.c:92-107 adds them to w[] from 0 (same re-layout as upload_jad_as_csr for a whole matrix) */
LIS_INT lisi_split_jad_part(LIS_MATRIX A, int upper, LIS_INT **optr, LIS_INT **oidx, LIS_SCALAR **oval)
{
	LIS_MATRIX_CORE P = upper ? A->U : A->L;
	const LIS_INT n = A->n;
	LIS_INT *ptr = (LIS_INT *)zalloc((size_t)n + 1, sizeof(LIS_INT));
	LIS_INT *idx = (LIS_INT *)zalloc((size_t)P->nnz, sizeof(LIS_INT));
	LIS_SCALAR *val = (LIS_SCALAR *)zalloc((size_t)P->nnz, sizeof(LIS_SCALAR));
	LIS_INT *fill = (LIS_INT *)zalloc((size_t)n, sizeof(LIS_INT));
	if (!ptr || !idx || !val || !fill) { free(ptr); free(idx); free(val); free(fill); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", P->nnz); }
	for (LIS_INT j = 0; j < P->maxnzr; j++) for (LIS_INT s = 0; s < P->ptr[j + 1] - P->ptr[j]; s++) ptr[P->row[s] + 1]++;
	for (LIS_INT r = 0; r < n; r++) ptr[r + 1] += ptr[r];
	memcpy(fill, ptr, sizeof(LIS_INT) * (size_t)n);
	for (LIS_INT j = 0; j < P->maxnzr; j++) {
		const LIS_INT b = P->ptr[j], len = P->ptr[j + 1] - b;
		for (LIS_INT s = 0; s < len; s++) { const LIS_INT at = fill[P->row[s]]++; idx[at] = P->index[b + s]; val[at] = P->value[b + s]; }
	}
	free(fill);
	*optr = ptr; *oidx = idx; *oval = val;
	return LIS_SUCCESS;
}
