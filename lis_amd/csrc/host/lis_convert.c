/*
 * lis_convert.c -- lis_matrix_convert / lis_matrix_copy: the storage layouts of the six served formats.
 *
 * One-off host work that DEFINES the data layout each SpMV kernel streams, and with it the summation
 * order of every row.  Layouts are the reference's at one OpenMP thread (its DIA and JAD layouts are
 * blocked by the producer's thread count): csr2ell src/matrix/lis_matrix_ell.c:958-1070,
 * csr2dia lis_matrix_dia.c:1191-1304, csr2jad lis_matrix_jad.c:1591-1770, csr2bsr lis_matrix_bsr.c:351-552,
 * csr2csc lis_matrix_csc.c:904-1087; dispatcher lis_matrix_ops.c:128-322.
 */
#include "lis_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEW(p, T, count) do { (p) = (T *)malloc(sizeof(T) * (size_t)((count) > 0 ? (count) : 1)); \
	if (!(p)) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)(count)); goto fail; } } while (0)

static LIS_INT finish(LIS_MATRIX Aout, LIS_INT err)
{
	if (err) return err;
	err = lis_matrix_assemble(Aout);
	if (err) lisi_matrix_storage_destroy(Aout);
	return err;
}

/* ------------------------------------------------------------------ CSR -> X */
static LIS_INT csr2ell(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n;
	LIS_INT err = 0, maxnzr = 0, *index = NULL; LIS_SCALAR *value = NULL;
	#pragma omp parallel for reduction(max : maxnzr) num_threads(lisi_host_threads())
	for (LIS_INT i = 0; i < n; i++) if (A->ptr[i + 1] - A->ptr[i] > maxnzr) maxnzr = A->ptr[i + 1] - A->ptr[i];
	NEW(index, LIS_INT, (size_t)n * maxnzr); NEW(value, LIS_SCALAR, (size_t)n * maxnzr);
	/* slot j of every row, rows in chunks: each thread writes whole cache lines of the column-major arrays;
	 * padding: value 0 on the row's own column */
	#pragma omp parallel for schedule(static) num_threads(lisi_host_threads())
	for (LIS_INT c = 0; c < (n + 1023) / 1024; c++) {
		const LIS_INT i0 = c * 1024, i1 = i0 + 1024 < n ? i0 + 1024 : n;
		for (LIS_INT j = 0; j < maxnzr; j++)
			for (LIS_INT i = i0; i < i1; i++) {
				const LIS_INT k = A->ptr[i] + j;
				const int have = k < A->ptr[i + 1];
				value[(size_t)j * n + i] = have ? A->value[k] : 0.0;
				index[(size_t)j * n + i] = have ? A->index[k] : i;
			}
	}
	return finish(B, lis_matrix_set_ell(maxnzr, index, value, B));
fail:
	free(index); free(value);
	return err;
}

static LIS_INT csr2csc(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n, np = A->np, nnz = A->nnz;
	LIS_INT err = 0, *ptr = NULL, *index = NULL, *fill = NULL; LIS_SCALAR *value = NULL;
	NEW(ptr, LIS_INT, np + 1); NEW(index, LIS_INT, nnz); NEW(value, LIS_SCALAR, nnz); NEW(fill, LIS_INT, np + 1);
	memset(fill, 0, sizeof(LIS_INT) * (size_t)(np + 1));
	/* a counting sort by column that keeps the rows ascending inside each column.  Threads own column RANGES and each walks
	 * all rows in order, taking the entries of its columns: no shared counters, the same arrays as the serial sweep */
	int nth = 1;
	#ifdef _OPENMP
	nth = lisi_host_threads();
	if ((long long)nnz < 4000000) nth = 1;
	#endif
	#pragma omp parallel for schedule(static, 1) num_threads(nth)
	for (int t = 0; t < nth; t++) {
		const LIS_INT c0 = (LIS_INT)((long long)np * t / nth), c1 = (LIS_INT)((long long)np * (t + 1) / nth);
		for (LIS_INT k = 0; k < nnz; k++) { const LIS_INT c = A->index[k]; if (c >= c0 && c < c1) fill[c]++; }
	}
	ptr[0] = 0;
	for (LIS_INT c = 0; c < np; c++) { ptr[c + 1] = ptr[c] + fill[c]; fill[c] = ptr[c]; }
	#pragma omp parallel for schedule(static, 1) num_threads(nth)
	for (int t = 0; t < nth; t++) {
		const LIS_INT c0 = (LIS_INT)((long long)np * t / nth), c1 = (LIS_INT)((long long)np * (t + 1) / nth);
		for (LIS_INT i = 0; i < n; i++)                       /* rows ascending inside each column */
			for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) {
				const LIS_INT c = A->index[k];
				if (c >= c0 && c < c1) { const LIS_INT dst = fill[c]++; value[dst] = A->value[k]; index[dst] = i; }
			}
	}
	free(fill);
	return finish(B, lis_matrix_set_csc(nnz, ptr, index, value, B));
fail:
	free(ptr); free(index); free(value); free(fill);
	return err;
}

static LIS_INT csr2dia(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n, np = A->np;
	LIS_INT err = 0, nnd = 0, *index = NULL, *slot = NULL; LIS_SCALAR *value = NULL;
	unsigned char *used = NULL;
	/* like the reference (lis_matrix_dia.c:1217) the INPUT rows are put in ascending column order first */
	int moved = 0;
	#pragma omp parallel for schedule(dynamic, 4096) num_threads(lisi_host_threads()) reduction(| : moved)
	for (LIS_INT i = 0; i < n; i++) {
		int unsorted = 0;
		for (LIS_INT k = A->ptr[i] + 1; k < A->ptr[i + 1]; k++) unsorted |= A->index[k] < A->index[k - 1];
		if (unsorted) { lisi_sort_row(A->ptr[i], A->ptr[i + 1], A->index, A->value); moved = 1; }
	}
	A->is_sorted = LIS_TRUE;
	if (moved && MDEV(A)->ready) lisd_mat_free(A);            /* its HBM copy had the old order */
	/* the diagonals that occur, ascending: one flag per possible offset -(n-1) .. np-1 instead of a sort of all offsets */
	const size_t span = (size_t)n + (size_t)np;
	used = (unsigned char *)calloc(span ? span : 1, 1);
	if (!used) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)span); goto fail; }
	#pragma omp parallel for schedule(static) num_threads(lisi_host_threads())
	for (LIS_INT i = 0; i < n; i++)
		for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) {
			/* test first: a stencil has a handful of offsets, and 16 threads storing into the same cache line 10^8 times
			 * (measured: 0.4 s at 160^3) is what the test avoids; racing writers store the same 1 */
			const size_t o = (size_t)(A->index[k] - i + n);
			if (!used[o]) used[o] = 1;
		}
	for (size_t o = 0; o < span; o++) nnd += used[o];
	NEW(index, LIS_INT, nnd); NEW(value, LIS_SCALAR, (size_t)n * nnd); NEW(slot, LIS_INT, span);
	nnd = 0;
	for (size_t o = 0; o < span; o++) if (used[o]) { slot[o] = nnd; index[nnd++] = (LIS_INT)((long long)o - n); }
	#pragma omp parallel for schedule(static) num_threads(lisi_host_threads())
	for (LIS_INT c = 0; c < (n + 1023) / 1024; c++) {
		const LIS_INT i0 = c * 1024, i1 = i0 + 1024 < n ? i0 + 1024 : n;
		for (LIS_INT d = 0; d < nnd; d++) memset(value + (size_t)d * n + i0, 0, sizeof(LIS_SCALAR) * (size_t)(i1 - i0));
		for (LIS_INT i = i0; i < i1; i++)
			for (LIS_INT k = A->ptr[i]; k < A->ptr[i + 1]; k++) value[(size_t)slot[(size_t)(A->index[k] - i + n)] * n + i] = A->value[k];
	}
	free(used); free(slot);
	return finish(B, lis_matrix_set_dia(nnd, index, value, B));
fail:
	free(used); free(slot); free(index); free(value);
	return err;
}

/* descending sort of key[] carrying tag[] with the reference's own partition scheme (middle pivot parked at
 * the end, strict Hoare scans; src/system/lis_sort.c:249-276): it is unstable, and the JAD permutation of
 * equal-length rows is whatever this exact scheme leaves -- parity of A->row needs the same scheme */
void lisi_sortr_ii(LIS_INT lo, LIS_INT hi, LIS_INT *key, LIS_INT *tag)
{
	while (lo < hi) {
		const LIS_INT mid = (lo + hi) / 2, pv = key[mid];
		LIS_INT t;
		t = key[mid]; key[mid] = key[hi]; key[hi] = t;
		t = tag[mid]; tag[mid] = tag[hi]; tag[hi] = t;
		LIS_INT a = lo, b = hi;
		while (a <= b) {
			while (key[a] > pv) a++;
			while (key[b] < pv) b--;
			if (a <= b) {
				t = key[a]; key[a] = key[b]; key[b] = t;
				t = tag[a]; tag[a] = tag[b]; tag[b] = t;
				a++; b--;
			}
		}
		lisi_sortr_ii(lo, b, key, tag);
		lo = a;
	}
}

/* The same sort with its recursion spread over the host threads: the two sides of a partition are disjoint ranges, so whoever sorts them
 * and in whatever order, every exchange is the one the sequential scheme makes -- the permutation is the reference's, element for element.
 * 256^3 (16.7 M rows, two distinct lengths: every partition pass swaps): 1.2 s -> 0.15 s on 16 threads. */
static void sortr_tasks(LIS_INT lo, LIS_INT hi, LIS_INT *key, LIS_INT *tag)
{
	while (lo < hi) {
		if (hi - lo < (1 << 15)) { lisi_sortr_ii(lo, hi, key, tag); return; }
		const LIS_INT mid = (lo + hi) / 2, pv = key[mid];
		LIS_INT t;
		t = key[mid]; key[mid] = key[hi]; key[hi] = t;
		t = tag[mid]; tag[mid] = tag[hi]; tag[hi] = t;
		LIS_INT a = lo, b = hi;
		while (a <= b) {
			while (key[a] > pv) a++;
			while (key[b] < pv) b--;
			if (a <= b) {
				t = key[a]; key[a] = key[b]; key[b] = t;
				t = tag[a]; tag[a] = tag[b]; tag[b] = t;
				a++; b--;
			}
		}
		const LIS_INT llo = lo, lhi = b;
		#pragma omp task firstprivate(llo, lhi) shared(key, tag)
		sortr_tasks(llo, lhi, key, tag);
		lo = a;
	}
}

/* what makes a JAD matrix out of a CSR one besides the entries: the longest row, the rows in the reference's length-sorted order
 * (perm[n]) and the starts of the jagged diagonals (ptr[maxnzr + 1]: one chunk).  Both arrays are the caller's (malloc). */
LIS_INT lisi_jad_order(LIS_MATRIX A, LIS_INT *maxnzr_out, LIS_INT **perm_out, LIS_INT **ptr_out)
{
	const LIS_INT n = A->n;
	LIS_INT err = 0, maxnzr = 0, *len = NULL, *perm = NULL, *ptr = NULL;
	NEW(len, LIS_INT, n);
	#pragma omp parallel for reduction(max : maxnzr) num_threads(lisi_host_threads())
	for (LIS_INT i = 0; i < n; i++) { len[i] = A->ptr[i + 1] - A->ptr[i]; if (len[i] > maxnzr) maxnzr = len[i]; }
	NEW(perm, LIS_INT, n); NEW(ptr, LIS_INT, maxnzr + 1);
	memset(ptr, 0, sizeof(LIS_INT) * (size_t)(maxnzr + 1));
	{	/* ptr[j+1] = rows with more than j entries: a histogram of the lengths, summed from the top */
		LIS_INT *hist = (LIS_INT *)calloc((size_t)maxnzr + 2, sizeof(LIS_INT));
		if (!hist) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", maxnzr); goto fail; }
		for (LIS_INT i = 0; i < n; i++) { perm[i] = i; hist[len[i]]++; }
		LIS_INT longer = 0;
		for (LIS_INT j = maxnzr; j >= 1; j--) { longer += hist[j]; ptr[j] = longer; }
		free(hist);
	}
	#pragma omp parallel num_threads(lisi_host_threads())
	{
		#pragma omp single
		sortr_tasks(0, n - 1, len, perm);
	}
	for (LIS_INT j = 0; j < maxnzr; j++) ptr[j + 1] += ptr[j];
	free(len);
	*maxnzr_out = maxnzr; *perm_out = perm; *ptr_out = ptr;
	return LIS_SUCCESS;
fail:
	free(len); free(perm); free(ptr);
	return err;
}

static LIS_INT csr2jad(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n, nnz = A->nnz;
	LIS_INT err = 0, maxnzr = 0, *len = NULL, *perm = NULL, *ptr = NULL, *index = NULL; LIS_SCALAR *value = NULL;
	LISCHK(lisi_jad_order(A, &maxnzr, &perm, &ptr));
	NEW(index, LIS_INT, nnz); NEW(value, LIS_SCALAR, nnz);
	#pragma omp parallel for schedule(static) num_threads(lisi_host_threads())
	for (LIS_INT s = 0; s < n; s++) {                          /* jagged diagonal j holds the j-th entry of every row long enough */
		const LIS_INT src = A->ptr[perm[s]], cnt = A->ptr[perm[s] + 1] - src;
		for (LIS_INT j = 0; j < cnt; j++) { value[ptr[j] + s] = A->value[src + j]; index[ptr[j] + s] = A->index[src + j]; }
	}
	free(len);
	return finish(B, lis_matrix_set_jad(nnz, maxnzr, perm, ptr, index, value, B));
fail:
	free(len); free(perm); free(ptr); free(index); free(value);
	return err;
}

static LIS_INT csr2bsr(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n, np = A->np, bnr = B->conv_bnr, bnc = B->conv_bnc, bs = bnr * bnc;
	const LIS_INT nr = 1 + (n - 1) / bnr, pad = (bnc - n % bnc) % bnc;
	const LIS_INT nc = (n == np) ? 1 + (n - 1) / bnc : 2 + (n - 1) / bnc + (pad + np - n - 1) / bnc;
	LIS_INT err = 0, *bptr = NULL, *bindex = NULL, *slot = NULL, *seen = NULL; LIS_SCALAR *value = NULL;
	NEW(bptr, LIS_INT, nr + 1);
	(void)nc;
#define BCOL(c) (((c) < n ? (c) : (c) + pad) / bnc)             /* ghost columns start on a fresh block (ref :425-428) */
#define BOFF(c) (((c) < n ? (c) : (c) + pad) % bnc)
	/* Block rows are independent.  The distinct block columns of one block row, in first-seen order, are few (a stencil: 7-14):
	 * a short list searched linearly from its end (neighbouring entries repeat the last block) replaces a table over all nc. */
	enum { LOCAL = 512 };
	bptr[0] = 0;
	int oom = 0;                                                /* a failed allocation inside the parallel loops: reported after them */
	#pragma omp parallel for schedule(dynamic, 1024) num_threads(lisi_host_threads())
	for (LIS_INT br = 0; br < nr; br++) {                       /* pass 1: distinct block columns per block row */
		LIS_INT list[LOCAL], nseen = 0, *big = NULL, cap = LOCAL;
		LIS_INT *cur = list;
		if (oom) { bptr[br + 1] = 0; continue; }
		for (LIS_INT ii = 0; ii < bnr && br * bnr + ii < n; ii++)
			for (LIS_INT k = A->ptr[br * bnr + ii]; k < A->ptr[br * bnr + ii + 1]; k++) {
				const LIS_INT bc = BCOL(A->index[k]);
				LIS_INT s = nseen - 1;
				while (s >= 0 && cur[s] != bc) s--;
				if (s < 0) {
					if (nseen == cap) {                             /* a very wide block row: the list moves to the heap */
						LIS_INT *nb = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)cap * 2);
						if (!nb) {
							#pragma omp atomic write
							oom = 1;
							nseen = 0; ii = bnr; break;                  /* give the row up: the whole conversion fails below */
						}
						memcpy(nb, cur, sizeof(LIS_INT) * (size_t)cap);
						free(big); big = nb; cur = nb; cap *= 2;
					}
					cur[nseen++] = bc;
				}
			}
		free(big);
		bptr[br + 1] = nseen;
	}
	if (oom) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc: distinct block columns of a block row\n"); goto fail; }
	for (LIS_INT br = 0; br < nr; br++) bptr[br + 1] += bptr[br];
	const LIS_INT bnnz = bptr[nr];
	NEW(bindex, LIS_INT, bnnz); NEW(value, LIS_SCALAR, (size_t)bnnz * bs);
	#pragma omp parallel for schedule(dynamic, 1024) num_threads(lisi_host_threads())
	for (LIS_INT br = 0; br < nr; br++) {                       /* pass 2: blocks in first-seen order, column-major inside */
		const LIS_INT first = bptr[br];
		LIS_INT next = first;
		for (LIS_INT ii = 0; ii < bnr && br * bnr + ii < n; ii++)
			for (LIS_INT k = A->ptr[br * bnr + ii]; k < A->ptr[br * bnr + ii + 1]; k++) {
				const LIS_INT bc = BCOL(A->index[k]), jc = BOFF(A->index[k]);
				LIS_INT s = next - 1;
				while (s >= first && bindex[s] != bc) s--;
				if (s < first) {
					s = next++;
					bindex[s] = bc;
					for (LIS_INT z = 0; z < bs; z++) value[(size_t)s * bs + z] = 0.0;
				}
				value[(size_t)s * bs + (size_t)jc * bnr + ii] = A->value[k];
			}
	}
#undef BCOL
#undef BOFF
	free(slot); free(seen);
	err = lis_matrix_set_bsr(bnr, bnc, bnnz, bptr, bindex, value, B);
	if (!err) {                                                  /* ref lis_matrix_bsr.c:539-548 */
		B->pad_comm = pad;
		if (B->commtable) B->commtable->pad = pad;               /* the halo of a BSR matrix lands behind the front padding */
	}
	return finish(B, err);
fail:
	free(bptr); free(bindex); free(value); free(slot); free(seen);
	return err;
}

LIS_INT lisi_convert_csr_to(LIS_MATRIX Ain, LIS_MATRIX Aout)
{
	switch (Aout->matrix_type) {
	case LIS_MATRIX_CSC: return csr2csc(Ain, Aout);
	case LIS_MATRIX_ELL: return csr2ell(Ain, Aout);
	case LIS_MATRIX_DIA: return csr2dia(Ain, Aout);
	case LIS_MATRIX_JAD: return csr2jad(Ain, Aout);
	case LIS_MATRIX_BSR: return csr2bsr(Ain, Aout);
	default: return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", Aout->matrix_type);
	}
}

/* ------------------------------------------------------------------ X -> CSR
 * (ELL / DIA / BSR drop their explicit zeros, as the reference's ell2csr / dia2csr / bsr2csr do) */
typedef struct { LIS_INT n, nnz, *ptr, *index; LIS_SCALAR *value; } csr_out;

static LIS_INT csr_out_alloc(csr_out *c, LIS_INT n, const LIS_INT *count)
{
	c->n = n;
	c->ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(n + 1));
	if (!c->ptr) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", n + 1);
	c->ptr[0] = 0;
	for (LIS_INT i = 0; i < n; i++) c->ptr[i + 1] = c->ptr[i] + count[i];
	c->nnz = c->ptr[n];
	c->index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(c->nnz > 0 ? c->nnz : 1));
	c->value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(c->nnz > 0 ? c->nnz : 1));
	if (!c->index || !c->value) { free(c->ptr); free(c->index); free(c->value); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", c->nnz); }
	return LIS_SUCCESS;
}

LIS_INT lisi_convert_to_csr(LIS_MATRIX A, LIS_MATRIX B)
{
	const LIS_INT n = A->n;
	LIS_INT *count = (LIS_INT *)calloc((size_t)(n > 0 ? n : 1), sizeof(LIS_INT));
	if (!count) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", n);
	csr_out c = {0};
	LIS_INT err = 0;
	switch (A->matrix_type) {
	case LIS_MATRIX_CSC:
		for (LIS_INT k = 0; k < A->nnz; k++) count[A->index[k]]++;
		if ((err = csr_out_alloc(&c, n, count))) break;
		memset(count, 0, sizeof(LIS_INT) * (size_t)n);
		for (LIS_INT col = 0; col < A->np; col++)
			for (LIS_INT k = A->ptr[col]; k < A->ptr[col + 1]; k++) {
				const LIS_INT r = A->index[k], dst = c.ptr[r] + count[r]++;
				c.index[dst] = col; c.value[dst] = A->value[k];
			}
		break;
	case LIS_MATRIX_ELL:
		for (LIS_INT j = 0; j < A->maxnzr; j++) for (LIS_INT i = 0; i < n; i++) if (A->value[(size_t)j * n + i] != 0.0) count[i]++;
		if ((err = csr_out_alloc(&c, n, count))) break;
		memset(count, 0, sizeof(LIS_INT) * (size_t)n);
		for (LIS_INT j = 0; j < A->maxnzr; j++) for (LIS_INT i = 0; i < n; i++) if (A->value[(size_t)j * n + i] != 0.0) {
			const LIS_INT dst = c.ptr[i] + count[i]++;
			c.index[dst] = A->index[(size_t)j * n + i]; c.value[dst] = A->value[(size_t)j * n + i];
		}
		break;
	case LIS_MATRIX_DIA:
		for (LIS_INT d = 0; d < A->nnd; d++) for (LIS_INT i = 0; i < n; i++) {
			const LIS_INT col = i + A->index[d];
			if (col >= 0 && col < A->np && A->value[(size_t)d * n + i] != 0.0) count[i]++;
		}
		if ((err = csr_out_alloc(&c, n, count))) break;
		memset(count, 0, sizeof(LIS_INT) * (size_t)n);
		for (LIS_INT d = 0; d < A->nnd; d++) for (LIS_INT i = 0; i < n; i++) {
			const LIS_INT col = i + A->index[d];
			if (col >= 0 && col < A->np && A->value[(size_t)d * n + i] != 0.0) {
				const LIS_INT dst = c.ptr[i] + count[i]++;
				c.index[dst] = col; c.value[dst] = A->value[(size_t)d * n + i];
			}
		}
		break;
	case LIS_MATRIX_JAD:
		for (LIS_INT j = 0; j < A->maxnzr; j++) for (LIS_INT s = 0; s < A->ptr[j + 1] - A->ptr[j]; s++) count[A->row[s]]++;
		if ((err = csr_out_alloc(&c, n, count))) break;
		memset(count, 0, sizeof(LIS_INT) * (size_t)n);
		for (LIS_INT j = 0; j < A->maxnzr; j++) for (LIS_INT s = 0; s < A->ptr[j + 1] - A->ptr[j]; s++) {
			const LIS_INT r = A->row[s], dst = c.ptr[r] + count[r]++;
			c.index[dst] = A->index[A->ptr[j] + s]; c.value[dst] = A->value[A->ptr[j] + s];
		}
		break;
	case LIS_MATRIX_BSR: {
		const LIS_INT bnr = A->bnr, bnc = A->bnc, bs = bnr * bnc;
		const LIS_INT pad = (bnc - n % bnc) % bnc;
		for (int pass = 0; pass < 2 && !err; pass++) {
			if (pass == 1) { if ((err = csr_out_alloc(&c, n, count))) break; memset(count, 0, sizeof(LIS_INT) * (size_t)n); }
			for (LIS_INT br = 0; br < A->nr; br++)
				for (LIS_INT b = A->bptr[br]; b < A->bptr[br + 1]; b++)
					for (LIS_INT jc = 0; jc < bnc; jc++) for (LIS_INT ii = 0; ii < bnr; ii++) {
						const LIS_INT r = br * bnr + ii;
						const LIS_SCALAR v = A->value[(size_t)b * bs + (size_t)jc * bnr + ii];
						if (r >= n || v == 0.0) continue;
						LIS_INT col = A->bindex[b] * bnc + jc;
						if (col >= n + pad) col -= pad; else if (col >= n) continue;
						if (pass == 0) count[r]++;
						else { const LIS_INT dst = c.ptr[r] + count[r]++; c.index[dst] = col; c.value[dst] = v; }
					}
		}
		/* a row's entries arrive block by block in stored block order, ascending column inside a block: the order
		 * lis_matrix_convert_bsr2csr leaves (lis_matrix_bsr.c convert loop: bj outer, j inner), unsorted blocks included */
		break; }
	default:
		err = LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
	free(count);
	if (err) return err;
	B->pad = 0; B->pad_comm = 0;                                 /* ref lis_matrix_bsr.c:673-683: a CSR matrix has no padding, */
	if (B->commtable) B->commtable->pad = 0;                     /* whatever the (duplicated) tables of the BSR source said    */
	return finish(B, lis_matrix_set_csr(c.nnz, c.ptr, c.index, c.value, B));
}

/* ------------------------------------------------------------------ same-type deep copy */
#define DUP(dst, src, T, count) do { if (src) { (dst) = (T *)malloc(sizeof(T) * (size_t)((count) > 0 ? (count) : 1)); \
	if (!(dst)) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)(count)); memcpy((dst), (src), sizeof(T) * (size_t)(count)); } } while (0)

LIS_INT lisi_matrix_deep_copy(LIS_MATRIX A, LIS_MATRIX B)
{
	const size_t n = (size_t)A->n;
	LIS_INT *ptr = NULL, *index = NULL, *row = NULL, *bptr = NULL, *bindex = NULL; LIS_SCALAR *value = NULL;
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		DUP(ptr, A->ptr, LIS_INT, n + 1); DUP(index, A->index, LIS_INT, A->nnz); DUP(value, A->value, LIS_SCALAR, A->nnz);
		return finish(B, lis_matrix_set_csr(A->nnz, ptr, index, value, B));
	case LIS_MATRIX_CSC:
		DUP(ptr, A->ptr, LIS_INT, (size_t)A->np + 1); DUP(index, A->index, LIS_INT, A->nnz); DUP(value, A->value, LIS_SCALAR, A->nnz);
		return finish(B, lis_matrix_set_csc(A->nnz, ptr, index, value, B));
	case LIS_MATRIX_ELL:
		DUP(index, A->index, LIS_INT, n * A->maxnzr); DUP(value, A->value, LIS_SCALAR, n * A->maxnzr);
		return finish(B, lis_matrix_set_ell(A->maxnzr, index, value, B));
	case LIS_MATRIX_DIA:
		DUP(index, A->index, LIS_INT, A->nnd); DUP(value, A->value, LIS_SCALAR, n * A->nnd);
		return finish(B, lis_matrix_set_dia(A->nnd, index, value, B));
	case LIS_MATRIX_JAD:
		DUP(row, A->row, LIS_INT, n); DUP(ptr, A->ptr, LIS_INT, (size_t)A->maxnzr + 1);
		DUP(index, A->index, LIS_INT, A->nnz); DUP(value, A->value, LIS_SCALAR, A->nnz);
		return finish(B, lis_matrix_set_jad(A->nnz, A->maxnzr, row, ptr, index, value, B));
	case LIS_MATRIX_BSR:
		DUP(bptr, A->bptr, LIS_INT, (size_t)A->nr + 1); DUP(bindex, A->bindex, LIS_INT, A->bnnz);
		DUP(value, A->value, LIS_SCALAR, (size_t)A->bnnz * A->bnr * A->bnc);
		return finish(B, lis_matrix_set_bsr(A->bnr, A->bnc, A->bnnz, bptr, bindex, value, B));
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
}

LIS_INT lis_matrix_copy(LIS_MATRIX Ain, LIS_MATRIX Aout)
{
	LISCHK(lisi_matrix_check(Ain, LISI_CHECK_ASSEMBLED));
	LISCHK(lisi_matrix_check(Aout, LISI_CHECK_NULL));
	if (MDEV(Ain)->device_only) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "matrix lives in HBM only\n");
	LISCHK(lis_matrix_merge(Ain));         /* a split matrix is copied in its merged form: the parts may hold a scaled system (lis_split.c) */
	LISCHK(lisp_fill_matrix(Ain));
	return lisi_matrix_deep_copy(Ain, Aout);
}

/* ------------------------------------------------------------------ dispatcher (ref lis_matrix_ops.c:128-322) */
static LIS_INT convert_impl(LIS_MATRIX Ain, LIS_MATRIX Aout);
LIS_INT lis_matrix_convert(LIS_MATRIX Ain, LIS_MATRIX Aout)
{
	LISCHK(convert_impl(Ain, Aout));
	lisd_mat_eager(Aout);                  /* resident mode: the upload belongs to the conversion, not to the first product */
	return LIS_SUCCESS;
}

static LIS_INT convert_impl(LIS_MATRIX Ain, LIS_MATRIX Aout)
{
	LISCHK(lisi_matrix_check(Ain, LISI_CHECK_ASSEMBLED));
	LISCHK(lisi_matrix_check(Aout, LISI_CHECK_NULL));
	const LIS_INT want = Aout->matrix_type;
	if (MDEV(Ain)->device_only) {          /* born in HBM: the conversions that are built there (ELL, DIA, CSC, BSR) are served, the host routines have nothing to read */
		int done = 0;
		if (Ain->matrix_type == LIS_MATRIX_CSR && want != LIS_MATRIX_CSR) LISCHK(lisd_convert_csr(Ain, Aout, &done));
		if (done) return LIS_SUCCESS;
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "matrix lives in HBM only: this conversion runs on host arrays -- convert the host matrix before uploading\n");
	}
	LISCHK(lis_matrix_merge(Ain));         /* ref lis_matrix_ops.c:142: a split input is merged first */
	if (Ain->matrix_type == want && !Ain->is_block) { LISCHK(lisp_fill_matrix(Ain)); return lisi_matrix_deep_copy(Ain, Aout); }
	if (Ain->matrix_type == LIS_MATRIX_CSR) {
		int done = 0;                          /* in HBM when Ain lives there (lis_device.c), else on the host arrays */
		LISCHK(lisd_convert_csr(Ain, Aout, &done));
		if (done) return LIS_SUCCESS;
		LISCHK(lisp_fill_matrix(Ain));
		return lisi_convert_csr_to(Ain, Aout);
	}
	LISCHK(lisp_fill_matrix(Ain));         /* the host routines below read Ain's arrays from several threads: they come home first */
	if (want == LIS_MATRIX_CSR) return lisi_convert_to_csr(Ain, Aout);
	LIS_MATRIX tmp;                                            /* X -> CSR -> Y */
	LISCHK(lis_matrix_duplicate(Ain, &tmp));
	LIS_INT err = lisi_convert_to_csr(Ain, tmp);
	if (!err) err = lisi_convert_csr_to(tmp, Aout);
	lis_matrix_destroy(tmp);
	return err;
}
