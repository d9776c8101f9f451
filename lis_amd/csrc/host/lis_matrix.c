/*
 * lis_matrix.c -- LIS_MATRIX lifecycle: create / set_size / set_<fmt> / assemble / duplicate / destroy.
 *
 * The state machine drivers depend on is the reference's (src/matrix/lis_matrix.c):
 *   create -> DECIDING_SIZE(-256) -> set_size -> NULL(-257) -> set_<fmt> -> -<FMT> -> assemble -> +<FMT>
 * Arrays handed to lis_matrix_set_<fmt> are adopted, not copied (src/matrix/lis_matrix_csr.c:98-103),
 * and freed by lis_matrix_destroy when is_destroy (default TRUE, lis_matrix.c:85,387-396).
 * Element-wise assembly (lis_matrix_set_value, lis_matrix.c:700) gathers rows on the host and becomes
 * CSR (or the type chosen with set_type) inside lis_matrix_assemble.
 */
#include "lis_internal.h"

/* rows under element-wise assembly live in the private tail, not in the public w_* fields */
typedef struct { LIS_INT len, cap; LIS_INT *col; LIS_SCALAR *val; } asm_row;
typedef struct { lisi_matrix m; asm_row *rows; } lisi_matrix_asm;   /* allocation unit of every matrix */

static asm_row **asm_rows(LIS_MATRIX A) { return &((lisi_matrix_asm *)A)->rows; }

static void asm_free(LIS_MATRIX A)
{
	asm_row *r = *asm_rows(A);
	if (!r) return;
	for (LIS_INT i = 0; i < A->n; i++) { free(r[i].col); free(r[i].val); }
	free(r);
	*asm_rows(A) = NULL;
}

static LIS_INT mat_alloc(LIS_MATRIX *out)
{
	lisi_matrix_asm *m = (lisi_matrix_asm *)calloc(1, sizeof(lisi_matrix_asm));
	if (!m) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)sizeof(lisi_matrix_asm));
	LIS_MATRIX A = &m->m.pub;
	A->label = LIS_LABEL_MATRIX;
	A->matrix_type = LIS_MATRIX_CSR;
	A->status = LIS_MATRIX_DECIDING_SIZE;
	A->w_annz = 10;
	A->conv_bnr = 2;
	A->conv_bnc = 2;
	A->is_destroy = LIS_TRUE;
	lisi_register(A, LISI_KIND_MATRIX);
	*out = A;
	return LIS_SUCCESS;
}

LIS_INT lisi_matrix_check(LIS_MATRIX A, int level)
{
	if (!lisi_is_registered(A)) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is undefined\n");
	if (level == LISI_CHECK_NULL) return LIS_SUCCESS;
	if (level == LISI_CHECK_NOT_ASSEMBLED) {
		if (A->status != LIS_MATRIX_DECIDING_SIZE && A->status != LIS_MATRIX_NULL && A->status != LIS_MATRIX_ASSEMBLING)
			return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A has already been assembled\n");
		return LIS_SUCCESS;
	}
	if (A->status == LIS_MATRIX_DECIDING_SIZE) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix size is undefined\n");
	if (level == LISI_CHECK_SIZE) return LIS_SUCCESS;
	if (A->status == LIS_MATRIX_NULL && A->n > 0) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix type is undefined\n");
	if (A->status <= LIS_MATRIX_ASSEMBLING && A->n > 0) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is assembling\n");
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_create(LIS_Comm comm, LIS_MATRIX *Amat)
{
	*Amat = NULL;
	LISCHK(mat_alloc(Amat));
	(*Amat)->comm = comm;
	(*Amat)->nprocs = lisg.nprocs ? lisg.nprocs : 1;
	(*Amat)->my_rank = lisg.rank;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_size(LIS_MATRIX A, LIS_INT local_n, LIS_INT global_n)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NULL));
	if (global_n > 0 && local_n > global_n)
		return LISI_ERR(LIS_ERR_ILL_ARG, "local n(=%D) is larger than global n(=%D)\n", local_n, global_n);
	if (local_n < 0 || global_n < 0)
		return LISI_ERR(LIS_ERR_ILL_ARG, "local n(=%D) or global n(=%D) are less than 0\n", local_n, global_n);
	if (local_n == 0 && global_n == 0)
		return LISI_ERR(LIS_ERR_ILL_ARG, "local n(=%D) and global n(=%D) are 0\n", local_n, global_n);
	if (lisg.nprocs > 1 && global_n > 0 && global_n < lisg.nprocs)
		return LISI_ERR(LIS_ERR_ILL_ARG, "global n(=%D) is smaller than nprocs(=%D)\n", global_n, lisg.nprocs);
	LIS_INT *ranges, is, ie, nprocs, my_rank;
	LISCHK(lisc_ranges_create(A->comm, &local_n, &global_n, &ranges, &is, &ie, &nprocs, &my_rank));
	A->status = LIS_MATRIX_NULL;
	A->ranges = ranges;
	A->n = local_n; A->gn = global_n; A->np = local_n;
	A->my_rank = my_rank; A->nprocs = nprocs;
	A->is = is; A->ie = ie;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_get_size(LIS_MATRIX A, LIS_INT *local_n, LIS_INT *global_n)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_SIZE));
	*local_n = A->n; *global_n = A->gn;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_get_range(LIS_MATRIX A, LIS_INT *is, LIS_INT *ie)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_SIZE));
	*is = A->is; *ie = A->ie;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_get_nnz(LIS_MATRIX A, LIS_INT *nnz)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_SIZE));
	*nnz = A->nnz;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_type(LIS_MATRIX A, LIS_INT matrix_type)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NOT_ASSEMBLED));
	if (matrix_type < LIS_MATRIX_CSR || matrix_type > LIS_MATRIX_DNS)
		return LISI_ERR(LIS_ERR_ILL_ARG, "matrix_type is %D (Set between 1 to %D)\n", matrix_type, LIS_MATRIX_DNS);
	A->matrix_type = matrix_type;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_get_type(LIS_MATRIX A, LIS_INT *matrix_type)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NULL));
	*matrix_type = A->matrix_type;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_destroyflag(LIS_MATRIX A, LIS_INT flag)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NULL));
	A->is_destroy = flag ? LIS_TRUE : LIS_FALSE;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_blocksize(LIS_MATRIX A, LIS_INT bnr, LIS_INT bnc, LIS_INT row[], LIS_INT col[])
{	/* ref lis_matrix.c:1078: fixed block sizes for the BSR conversion (row/col: VBR only, not served) */
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NULL));
	if (bnr <= 0 || bnc <= 0) return LISI_ERR(LIS_ERR_ILL_ARG, "bnr=%D <= 0 or bnc=%D <= 0\n", bnr, bnc);
	if (row || col) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "variable block sizes (VBR) are not served\n");
	A->conv_bnr = bnr; A->conv_bnc = bnc;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_is_assembled(LIS_MATRIX A) { return A->status != LIS_MATRIX_NULL ? !LIS_SUCCESS : LIS_SUCCESS; }

/* ------------------------------------------------------------------ array allocation + adoption */
/* The arrays of lis_matrix_malloc_<fmt> live on pages of the library's own (lis_pages.c, lisp_alloc_tracked): once a matrix has adopted them and its HBM copy is
 * built, a host write to them is SEEN (one page fault) and the copy is rebuilt before the next product -- the reference reads adopted arrays live on every call
 * (lis_matrix_csr.c:98-103, lis_matvec_csr.c:97-109).  Released by lis_matrix_destroy / lis_free, never by free().  Without a memory file: plain malloc. */
/* WHAT THIS CHANGES FOR A PROGRAM: the reference hands out lis_malloc memory, which is malloc memory.  Arrays on these pages (1) must not be given to
 * free() -- lis_free() / lis_matrix_destroy() release them, as they release the reference's; (2) are READ-ONLY while the adopting matrix has an HBM copy: a store into
 * them faults once and is then seen, but a system call that writes into them (fread / read / MPI_Recv into A->value) fails with EFAULT -- call
 * lis_amd_matrix_host_modified(A) first, which opens the pages; lis_matrix_unset() opens them too.  A program that needs plain malloc memory -- it frees the arrays
 * itself, or reads files into them between solves -- runs with LIS_AMD_PLAIN_MALLOC=1 / lis_amd_set_matrix_pages(0): lis_matrix_malloc_<fmt> then returns malloc
 * memory, nothing is protected, and lis_amd_matrix_host_modified(A) after an in-place edit is the contract (as for every array the caller malloc'ed itself). */
static void *matrix_array(size_t bytes)
{
	void *p = lisg.plain_malloc ? NULL : lisp_alloc_tracked(bytes);
	return p ? p : malloc(bytes ? bytes : 1);
}
LIS_INT lis_amd_set_matrix_pages(LIS_INT on) { lisg.plain_malloc = on ? 0 : 1; return LIS_SUCCESS; }
#define ALLOC_OR_FAIL(p, T, count) do { (p) = (T *)matrix_array(sizeof(T) * (size_t)((count) > 0 ? (count) : 1)); \
	if (!(p)) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)(count)); } while (0)

LIS_INT lis_matrix_malloc_csr(LIS_INT n, LIS_INT nnz, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value)
{
	*ptr = NULL; *index = NULL; *value = NULL;
	ALLOC_OR_FAIL(*ptr, LIS_INT, n + 1); ALLOC_OR_FAIL(*index, LIS_INT, nnz); ALLOC_OR_FAIL(*value, LIS_SCALAR, nnz);
	return LIS_SUCCESS;
}
LIS_INT lis_matrix_malloc_csc(LIS_INT n, LIS_INT nnz, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value)
{ return lis_matrix_malloc_csr(n, nnz, ptr, index, value); }
LIS_INT lis_matrix_malloc_bsr(LIS_INT n, LIS_INT bnr, LIS_INT bnc, LIS_INT bnnz, LIS_INT **bptr, LIS_INT **bindex, LIS_SCALAR **value)
{
	const LIS_INT nr = 1 + (n - 1) / bnr;
	*bptr = NULL; *bindex = NULL; *value = NULL;
	ALLOC_OR_FAIL(*bptr, LIS_INT, nr + 1); ALLOC_OR_FAIL(*bindex, LIS_INT, bnnz);
	ALLOC_OR_FAIL(*value, LIS_SCALAR, (size_t)bnnz * bnr * bnc);
	return LIS_SUCCESS;
}
LIS_INT lis_matrix_malloc_ell(LIS_INT n, LIS_INT maxnzr, LIS_INT **index, LIS_SCALAR **value)
{
	*index = NULL; *value = NULL;
	ALLOC_OR_FAIL(*index, LIS_INT, (size_t)n * maxnzr); ALLOC_OR_FAIL(*value, LIS_SCALAR, (size_t)n * maxnzr);
	return LIS_SUCCESS;
}
LIS_INT lis_matrix_malloc_dia(LIS_INT n, LIS_INT nnd, LIS_INT **index, LIS_SCALAR **value)
{
	*index = NULL; *value = NULL;
	ALLOC_OR_FAIL(*index, LIS_INT, nnd); ALLOC_OR_FAIL(*value, LIS_SCALAR, (size_t)n * nnd);
	return LIS_SUCCESS;
}
LIS_INT lis_matrix_malloc_jad(LIS_INT n, LIS_INT nnz, LIS_INT maxnzr, LIS_INT **perm, LIS_INT **ptr, LIS_INT **index, LIS_SCALAR **value)
{	/* one chunk: ptr has maxnzr+1 entries (the reference sizes it nthreads*(maxnzr+1), lis_matrix_jad.c:1513) */
	*perm = NULL; *ptr = NULL; *index = NULL; *value = NULL;
	ALLOC_OR_FAIL(*perm, LIS_INT, n); ALLOC_OR_FAIL(*ptr, LIS_INT, maxnzr + 1);
	ALLOC_OR_FAIL(*index, LIS_INT, nnz); ALLOC_OR_FAIL(*value, LIS_SCALAR, nnz);
	return LIS_SUCCESS;
}

/* the reference's quirk: set_<fmt> on a matrix whose status is not NULL returns success WITHOUT
 * adopting the arrays (lis_matrix_csr.c:88-91) */
#define SET_GUARD(A) do { if (lis_matrix_is_assembled(A)) return LIS_SUCCESS; \
	if (!lisi_is_registered(A)) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is undefined\n"); } while (0)

LIS_INT lis_matrix_set_csr(LIS_INT nnz, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->ptr = ptr; A->index = index; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_CSR; A->nnz = nnz;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_csc(LIS_INT nnz, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->ptr = ptr; A->index = index; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_CSC; A->nnz = nnz;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_bsr(LIS_INT bnr, LIS_INT bnc, LIS_INT bnnz, LIS_INT *bptr, LIS_INT *bindex, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->bptr = bptr; A->bindex = bindex; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_BSR; A->is_block = LIS_TRUE;
	A->bnnz = bnnz;
	A->nr = (A->n - 1) / bnr + 1;
	if (A->n == A->np) {                      /* ref lis_matrix_bsr.c:95-104: pad so vectors cover whole blocks */
		A->nc = 1 + (A->n - 1) / bnc;
		A->pad = (bnc - A->n % bnc) % bnc;
	} else {
		A->nc = 2 + (A->n - 1) / bnc + (A->np - A->n - 1) / bnc;
		A->pad = (bnc - A->n % bnc) % bnc + (bnc - (A->np - A->n) % bnc) % bnc;
	}
	A->bnr = bnr; A->bnc = bnc;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_ell(LIS_INT maxnzr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->index = index; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_ELL; A->maxnzr = maxnzr;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_dia(LIS_INT nnd, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->index = index; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_DIA; A->nnd = nnd;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_jad(LIS_INT nnz, LIS_INT maxnzr, LIS_INT *perm, LIS_INT *ptr, LIS_INT *index, LIS_SCALAR *value, LIS_MATRIX A)
{
	SET_GUARD(A);
	A->row = perm; A->ptr = ptr; A->index = index; A->value = value;
	A->is_copy = LIS_FALSE; A->status = -LIS_MATRIX_JAD; A->nnz = nnz; A->maxnzr = maxnzr;
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ element-wise assembly */
LIS_INT lis_matrix_set_value(LIS_INT flag, LIS_INT i, LIS_INT j, LIS_SCALAR value, LIS_MATRIX A)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NOT_ASSEMBLED));
	if (A->status == LIS_MATRIX_DECIDING_SIZE) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix size is undefined\n");
	if (A->origin) { i--; j--; }
	if (i < 0 || j < 0) return LISI_ERR(LIS_ERR_ILL_ARG, "i(=%D) or j(=%D) are less than 0\n", i, j);
	if (i >= A->gn || j >= A->gn) return LISI_ERR(LIS_ERR_ILL_ARG, "i(=%D) or j(=%D) are larger than global n=(%D)\n", i, j, A->gn);
	if (i < A->is || i >= A->ie) return LIS_SUCCESS;          /* other ranks' rows are ignored, as in the reference */
	asm_row **rows = asm_rows(A);
	if (!*rows) {
		*rows = (asm_row *)calloc((size_t)(A->n > 0 ? A->n : 1), sizeof(asm_row));
		if (!*rows) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", A->n);
		A->status = LIS_MATRIX_ASSEMBLING;
		A->is_copy = LIS_TRUE;
	}
	asm_row *r = &(*rows)[i - A->is];
	for (LIS_INT k = 0; k < r->len; k++)
		if (r->col[k] == j) {
			if (flag == LIS_INS_VALUE) r->val[k] = value; else r->val[k] += value;
			return LIS_SUCCESS;
		}
	if (r->len == r->cap) {
		r->cap = r->cap ? 2 * r->cap : (A->w_nnz && A->w_nnz[i - A->is] > 0 ? A->w_nnz[i - A->is] : A->w_annz);
		r->col = (LIS_INT *)realloc(r->col, sizeof(LIS_INT) * (size_t)r->cap);
		r->val = (LIS_SCALAR *)realloc(r->val, sizeof(LIS_SCALAR) * (size_t)r->cap);
		if (!r->col || !r->val) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", r->cap);
	}
	r->col[r->len] = j; r->val[r->len] = value; r->len++;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_set_values(LIS_INT flag, LIS_INT n, LIS_SCALAR value[], LIS_MATRIX A)
{	/* a dense n x n block, row-major, through lis_matrix_set_value; errors of the single calls are dropped as in the
	 * reference (lis_matrix.c:808-822) */
	for (LIS_INT i = 0; i < n; i++)
		for (LIS_INT j = 0; j < n; j++) (void)lis_matrix_set_value(flag, i, j, value[(size_t)i * n + j], A);
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_malloc(LIS_MATRIX A, LIS_INT nnz_row, LIS_INT nnz[])
{	/* expected entries per row for the element-wise assembly (lis_matrix.c:592-625): a capacity hint here */
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NOT_ASSEMBLED));
	const LIS_INT n = A->n;
	if (!A->w_nnz) {
		A->w_nnz = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(n > 0 ? n : 1));
		if (!A->w_nnz) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", n);
	}
	if (nnz == NULL) { A->w_annz = nnz_row > 0 ? nnz_row : 1; for (LIS_INT k = 0; k < n; k++) A->w_nnz[k] = nnz_row; }
	else for (LIS_INT k = 0; k < n; k++) A->w_nnz[k] = nnz[k];
	return LIS_SUCCESS;
}

static LIS_INT assemble_rows_to_csr(LIS_MATRIX A)
{
	asm_row *rows = *asm_rows(A);
	const LIS_INT n = A->n;
	LIS_INT nnz = 0;
	for (LIS_INT i = 0; i < n; i++) nnz += rows[i].len;
	LIS_INT *ptr, *index; LIS_SCALAR *value;
	LISCHK(lis_matrix_malloc_csr(n, nnz, &ptr, &index, &value));
	ptr[0] = 0;
	for (LIS_INT i = 0; i < n; i++) {
		memcpy(index + ptr[i], rows[i].col, sizeof(LIS_INT) * (size_t)rows[i].len);
		memcpy(value + ptr[i], rows[i].val, sizeof(LIS_SCALAR) * (size_t)rows[i].len);
		ptr[i + 1] = ptr[i] + rows[i].len;
	}
	asm_free(A);
	A->ptr = ptr; A->index = index; A->value = value; A->nnz = nnz;
	return LIS_SUCCESS;
}

static LIS_INT assemble_impl(LIS_MATRIX A)
{
	LISCHK(lisi_matrix_check(A, LISI_CHECK_SIZE));
	if (A->status == LIS_MATRIX_NULL && A->n > 0) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix type is undefined\n");
	if (A->status == LIS_MATRIX_ASSEMBLING) {
		const LIS_INT want = A->matrix_type;
		LISCHK(assemble_rows_to_csr(A));
		A->status = LIS_MATRIX_CSR; A->matrix_type = LIS_MATRIX_CSR;
		if (lisg.nprocs > 1) { LISCHK(lisc_matrix_g2l(A)); LISCHK(lisc_commtable_create(A)); A->is_comm = LIS_TRUE; }
		return lisi_matrix_retype(A, want, 0);   /* ref lis_matrix.c:630-643: convert in place to the requested type */
	}
	if (A->status < 0 && A->n > 0) {
		A->status = -A->status;
		A->matrix_type = A->status;
		if (A->matrix_type == LIS_MATRIX_JAD && !A->work) {     /* ref lis_matrix.c:661-669 */
			A->work = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)A->n);
			if (!A->work) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", A->n);
		}
	}
	if (lisg.nprocs > 1 && !A->is_pmat && !MDEV(A)->device_only) {
		if (!A->l2g_map && A->matrix_type == LIS_MATRIX_CSR) LISCHK(lisc_matrix_g2l(A));
		if (!A->commtable && A->l2g_map) LISCHK(lisc_commtable_create(A));
	}
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_assemble(LIS_MATRIX A)
{
	LISCHK(assemble_impl(A));
	lisd_mat_eager(A);                     /* resident mode: upload + row split now, outside any timed loop of the caller */
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ duplicate / destroy */
LIS_INT lis_matrix_duplicate(LIS_MATRIX Ain, LIS_MATRIX *Aout)
{	/* a sized, unassembled matrix with the partition (ranges, l2g map, halo tables) of Ain: ref lis_matrix.c:458-590 */
	LISCHK(lisi_matrix_check(Ain, LISI_CHECK_ASSEMBLED));
	*Aout = NULL;
	LISCHK(mat_alloc(Aout));
	LIS_MATRIX B = *Aout;
	if (Ain->ranges) {
		B->ranges = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(Ain->nprocs + 1));
		memcpy(B->ranges, Ain->ranges, sizeof(LIS_INT) * (size_t)(Ain->nprocs + 1));
	}
	if (Ain->l2g_map && Ain->np > Ain->n) {
		B->l2g_map = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(Ain->np - Ain->n));
		memcpy(B->l2g_map, Ain->l2g_map, sizeof(LIS_INT) * (size_t)(Ain->np - Ain->n));
	}
	if (Ain->commtable) {
		LIS_COMMTABLE s = Ain->commtable, t = (LIS_COMMTABLE)calloc(1, sizeof(struct LIS_COMMTABLE_STRUCT));
		*t = *s;
		const LIS_INT nb = s->neibpetot;
		t->neibpe = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb > 0 ? nb : 1));
		t->import_ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb + 1));
		t->export_ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb + 1));
		t->import_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(s->imnnz > 0 ? s->imnnz : 1));
		t->export_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(s->exnnz > 0 ? s->exnnz : 1));
		memcpy(t->neibpe, s->neibpe, sizeof(LIS_INT) * (size_t)nb);
		memcpy(t->import_ptr, s->import_ptr, sizeof(LIS_INT) * (size_t)(nb + 1));
		memcpy(t->export_ptr, s->export_ptr, sizeof(LIS_INT) * (size_t)(nb + 1));
		memcpy(t->import_index, s->import_index, sizeof(LIS_INT) * (size_t)s->imnnz);
		memcpy(t->export_index, s->export_index, sizeof(LIS_INT) * (size_t)s->exnnz);
		t->ws = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(s->wssize > 0 ? s->wssize : 1));
		t->wr = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(s->wrsize > 0 ? s->wrsize : 1));
		B->commtable = t;
	}
	B->status = LIS_MATRIX_NULL;
	B->n = Ain->n; B->gn = Ain->gn; B->np = Ain->np;
	B->comm = Ain->comm; B->my_rank = Ain->my_rank; B->nprocs = Ain->nprocs;
	B->is = Ain->is; B->ie = Ain->ie; B->origin = Ain->origin;
	B->is_destroy = Ain->is_destroy;
	B->is_pmat = Ain->is_pmat; B->is_comm = Ain->is_comm;
	return LIS_SUCCESS;
}

LIS_INT lisi_matrix_storage_destroy(LIS_MATRIX A)
{
	lisi_matrix_dlu_destroy(A);
	asm_free(A);
	if (A->is_destroy) {
		LIS_INT *ia[] = {A->ptr, A->row, A->col, A->index, A->bptr, A->bindex};
		for (int k = 0; k < 6; k++) if (!lisp_free_array(ia[k])) free(ia[k]);         /* (arrays of a matrix converted in HBM live on pages of their own) */
		if (!lisp_free_array(A->value)) free(A->value);
		free(A->work);
		free(A->conv_row); free(A->conv_col);
	}
	lisd_mat_free(A);                       /* (after the arrays: one that was never read dies unmaterialised) */
	lisp_matrix_release(A, 1);              /* arrays that outlive the matrix (is_destroy off) are nobody's from here on */
	A->ptr = A->row = A->col = A->index = A->bptr = A->bindex = NULL;
	A->value = A->work = NULL;
	A->conv_row = A->conv_col = NULL;
	return LIS_SUCCESS;
}

/* move every field of src into dst (dst keeps its address); src is left hollow for free() */
LIS_INT lisi_matrix_copy_header(LIS_MATRIX src, LIS_MATRIX dst)
{
	free(dst->ranges); free(dst->l2g_map); free(dst->w_nnz);
	if (dst->commtable) lisc_commtable_destroy(dst->commtable);
	memcpy(dst, src, sizeof(struct LIS_MATRIX_STRUCT));
	lisp_reown(src, dst);                    /* arrays still held in HBM only belong to dst now */
	*MDEV(dst) = *MDEV(src);
	memset(MDEV(src), 0, sizeof(lisd_mat));
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_unset(LIS_MATRIX A)
{	/* ref lis_matrix.c:1110: drop the arrays without freeing them (the caller keeps ownership) */
	LISCHK(lisi_matrix_check(A, LISI_CHECK_NULL));
	if (A->is_copy) lisi_matrix_storage_destroy(A);
	lisd_mat_free(A);
	lisp_matrix_release(A, 1);              /* the caller's again */
	A->ptr = A->row = A->col = A->index = A->bptr = A->bindex = NULL;
	A->value = NULL;
	A->is_copy = LIS_FALSE;
	A->status = LIS_MATRIX_NULL;
	return LIS_SUCCESS;
}

LIS_INT lis_matrix_destroy(LIS_MATRIX A)
{
	if (A && lisi_is_registered(A)) {
		lisi_matrix_storage_destroy(A);
		free(A->w_nnz);
		free(A->l2g_map);
		if (A->commtable) lisc_commtable_destroy(A->commtable);
		free(A->ranges);
		lisi_unregister(A);
		free(A);
	}
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ diagonal */
LIS_INT lis_matrix_get_diagonal(LIS_MATRIX A, LIS_VECTOR D)
{	/* ref lis_matrix_ops.c:728 -> per-format get_diagonal; first stored entry with column == row, else 0 */
	LISCHK(lisi_matrix_check(A, LISI_CHECK_ASSEMBLED));
	if (A->n != D->n) return LISI_ERR(LIS_ERR_ILL_ARG, "length of diagonal D and row of A is not equal\n");
	lisd_mat *d = MDEV(A);
	if (d->device_only || (d->ready && d->type == LIS_MATRIX_CSR && A->matrix_type == LIS_MATRIX_CSR)) {
		double *dd;
		LISCHK(lisd_mat_ready(A));
		LISCHK(lisd_vec_out(D, &dd));
		HIPCHK(liship_csr_diagonal_f64(A->n, d->ptr, d->index, d->value, dd, lisg.stream));
		return lisd_vec_done(D);
	}
	/* an ELL / DIA matrix whose arrays live in HBM -- made there by lis_matrix_convert, its host arrays not asked for yet -- gives its diagonal there too: the
	 * host loop below would first bring 100 n / 56 n bytes home across PCIe for 8 n of result (Jacobi at 512^3: 13 GB).  Same entry, same value:
	 * the native arrays, or the row form, which lists the format's terms in the format's order (padding and explicit zeros included). */
	if (d->ready && !A->is_splited && lisg.nprocs == 1 && (A->matrix_type == LIS_MATRIX_ELL || A->matrix_type == LIS_MATRIX_DIA) &&
	    (d->type == A->matrix_type || (d->type == LIS_MATRIX_CSR && d->ptr && d->index && d->value)) && lisp_lazy_arrays(A) > 0) {
		double *dd;
		LISCHK(lisd_vec_out(D, &dd));
		if (d->type == LIS_MATRIX_CSR) HIPCHK(liship_csr_diagonal_f64(A->n, d->ptr, d->index, d->value, dd, lisg.stream));
		else if (d->type == LIS_MATRIX_ELL) HIPCHK(liship_ell_diagonal_f64(A->n, d->maxnzr, d->index, d->value, dd, lisg.stream));
		else HIPCHK(liship_dia_diagonal_f64(A->n, d->nnd, d->index, d->value, dd, lisg.stream));
		return lisd_vec_done(D);
	}
	const LIS_INT n = A->n;
	LISCHK(lisp_fill_matrix(A));
	LISCHK(lisd_vec_host_write(D, (size_t)(D->np + D->pad) > (size_t)n));   /* the host array is about to be written (entries beyond n keep what they hold) */
	LIS_SCALAR *out = D->value;
	for (LIS_INT i = 0; i < n; i++) out[i] = 0.0;
	if (A->is_splited) {                /* the split form holds the diagonal itself (ref lis_matrix_csr.c:532-545, lis_matrix_bsr.c get_diagonal) */
		if (A->matrix_type == LIS_MATRIX_BSR) {
			const size_t bs = (size_t)A->bnr * A->bnc;
			for (LIS_INT br = 0; br < A->nr; br++)
				for (LIS_INT j = 0; j < A->bnr && br * A->bnr + j < n; j++) out[br * A->bnr + j] = A->D->value[bs * br + (size_t)j * A->bnr + j];
		} else memcpy(out, A->D->value, sizeof(LIS_SCALAR) * (size_t)n);
		lis_amd_vector_host_modified(D);
		return LIS_SUCCESS;
	}
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		for (LIS_INT i = 0; i < n; i++)
			for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) if (A->index[k] == i) { out[i] = A->value[k]; break; }
		break;
	case LIS_MATRIX_CSC:
		for (LIS_INT i = 0; i < n; i++)
			for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) if (A->index[k] == i) { out[i] = A->value[k]; break; }
		break;
	case LIS_MATRIX_ELL:
		for (LIS_INT i = 0; i < n; i++)
			for (LIS_INT j = 0; j < A->maxnzr; j++) if (A->index[(size_t)j * n + i] == i) { out[i] = A->value[(size_t)j * n + i]; break; }
		break;
	case LIS_MATRIX_DIA:
		for (LIS_INT dgl = 0; dgl < A->nnd; dgl++) if (A->index[dgl] == 0) { memcpy(out, A->value + (size_t)dgl * n, sizeof(LIS_SCALAR) * (size_t)n); break; }
		break;
	case LIS_MATRIX_JAD:
		for (LIS_INT s = 0; s < n; s++)
			for (LIS_INT j = 0; j < A->maxnzr && s < A->ptr[j + 1] - A->ptr[j]; j++)
				if (A->index[A->ptr[j] + s] == A->row[s]) { out[A->row[s]] = A->value[A->ptr[j] + s]; break; }
		break;
	case LIS_MATRIX_BSR: {
		const LIS_INT bs = A->bnr * A->bnc;
		for (LIS_INT br = 0; br < A->nr; br++)
			for (LIS_INT b = A->bptr[br]; b < A->bptr[br + 1]; b++) {
				for (LIS_INT ii = 0; ii < A->bnr; ii++) {
					const LIS_INT r = br * A->bnr + ii, c = r - A->bindex[b] * A->bnc;
					if (r < n && c >= 0 && c < A->bnc) out[r] = A->value[(size_t)b * bs + (size_t)c * A->bnr + ii];
				}
			}
		break; }
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
	VDEV(D)->host_valid = 1; VDEV(D)->dev_valid = 0;
	return LIS_SUCCESS;
}
