/*
 * lis_io.c -- the data formats either side of the hot path: Matrix Market (+ Lis's extensions) in,
 * Matrix Market / plain / Lis-ASCII out.
 *
 * Behaviour follows the reference reader src/system/lis_input.c:67-171 (format sniffing),
 * lis_input_mm.c:62-146 (dispatch, conversion to the requested storage type), :326-411 (banner),
 * :413-458 (size line "nr nc nnz [isb isx [isbin]]"), :699-1069 (coordinate -> CSR) and :148-324
 * (the optional right-hand side / initial guess appended to the matrix file).  What fixes the BITS
 * of every later SpMV is the in-row entry order the reader produces:
 *     row r receives its entries in FILE order; for `symmetric` files the mirrored entry (c,r) of
 *     an off-diagonal line "r c v" is appended to row c BEFORE (r,c) is appended to row r.
 * The implementation is not the reference's two fgets/sscanf passes: the file is slurped once, every
 * line is parsed once (hand-rolled integer scan + strtod, which rounds like sscanf("%lg")) into
 * triplets, and a counting sort places them.  Every rank reads the whole file and keeps rows
 * [is,ie) with global column numbers, as the reference's MPI build does.
 */
#include <stdio.h>
#include <ctype.h>
#include <errno.h>
#include <omp.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include "lis_internal.h"

#define MM_BANNER "%%MatrixMarket"

typedef struct { char *buf; size_t len, pos; } slurp_t;

static int io_timing = -1;
static double io_t0;
static void io_mark(const char *what)
{
	if (io_timing < 0) { const char *e = getenv("LIS_AMD_IO_TIMING"); io_timing = (e && e[0] == '1'); io_t0 = lis_wtime(); }
	if (!io_timing) return;
	const double t = lis_wtime();
	if (what) fprintf(stderr, "lis_amd io: %-28s %8.3f s\n", what, t - io_t0);
	io_t0 = t;
}

static LIS_INT slurp_file(const char *path, slurp_t *s)
{
	io_mark(NULL);
	memset(s, 0, sizeof(*s));
	FILE *f = fopen(path, "rb");
	if (!f) return LISI_ERR(LIS_ERR_FILE_IO, "cannot open file %s\n", path);
	size_t cap = 1 << 16, len = 0;
	struct stat st;
	const int regular = fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
	if (regular) cap = (size_t)st.st_size + 1;
	char *buf = (char *)malloc(cap + 1);
	if (!buf) { fclose(f); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)cap); }
	if (regular && st.st_size > (16 << 20)) {
		/* a large regular file: the host threads read disjoint pieces (the first touch of the buffer's pages is most of the cost) */
		const size_t size = (size_t)st.st_size;
		const int T = lisi_host_threads(), fd = fileno(f);
		int bad = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(|:bad)
		for (int t = 0; t < T; t++) {
			size_t at = size / (size_t)T * (size_t)t;
			const size_t to = t == T - 1 ? size : size / (size_t)T * (size_t)(t + 1);
			while (at < to) {
				const ssize_t got = pread(fd, buf + at, to - at, (off_t)at);
				if (got <= 0) { bad = 1; break; }
				at += (size_t)got;
			}
		}
		if (bad) { free(buf); fclose(f); return LISI_ERR(LIS_ERR_FILE_IO, "cannot read file %s\n", path); }
		len = size;
	} else
	for (;;) {
		if (len == cap) {
			cap *= 2;
			char *nb = (char *)realloc(buf, cap + 1);
			if (!nb) { free(buf); fclose(f); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)cap); }
			buf = nb;
		}
		size_t got = fread(buf + len, 1, cap - len, f);
		if (got == 0) break;
		len += got;
	}
	fclose(f);
	buf[len] = '\0';
	s->buf = buf; s->len = len; s->pos = 0;
	io_mark("file read");
	return LIS_SUCCESS;
}

/* next text line [*b,*e) without its newline; the byte at *e is overwritten with NUL. 0 at end of file */
static int next_line(slurp_t *s, char **b, char **e)
{
	if (s->pos >= s->len) return 0;
	char *p = s->buf + s->pos;
	char *nl = (char *)memchr(p, '\n', s->len - s->pos);
	char *end = nl ? nl : s->buf + s->len;
	s->pos = (size_t)(end - s->buf) + (nl ? 1 : 0);
	*end = '\0';
	*b = p; *e = end;
	return 1;
}

static int scan_int(char **pp, long *out)
{	/* "%d": blanks, optional sign, digits */
	char *p = *pp;
	while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f') p++;
	int neg = 0;
	if (*p == '+' || *p == '-') { neg = (*p == '-'); p++; }
	if (*p < '0' || *p > '9') return 0;
	long v = 0;
	while (*p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
	*out = neg ? -v : v;
	*pp = p;
	return 1;
}

static int scan_double(char **pp, double *out)
{	/* "%lg" */
	char *end;
	double v = strtod(*pp, &end);
	if (end == *pp) return 0;
	*out = v; *pp = end;
	return 1;
}

static void lower(char *p) { for (; *p; p++) *p = (char)tolower((unsigned char)*p); }

/* ------------------------------------------------------------------ Matrix Market banner + size line */
typedef struct { int coordinate, symmetric; LIS_INT nr, nc, nnz, isb, isx, isbin; } mm_head;

static LIS_INT mm_banner(slurp_t *s, const char *object, mm_head *h)
{	/* ref lis_input_mm.c:326-411 */
	char *b, *e;
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	char banner[64] = "", mtx[64] = "", fmt[64] = "", dtype[64] = "", dstruct[64] = "";
	sscanf(b, "%63s %63s %63s %63s %63s", banner, mtx, fmt, dtype, dstruct);
	lower(mtx); lower(fmt); lower(dtype); lower(dstruct);
	if (strncmp(banner, MM_BANNER, strlen(MM_BANNER)) != 0 || strncmp(mtx, object, strlen(object)) != 0)
		return LISI_ERR(LIS_ERR_FILE_IO, "Not Matrix Market banner\n");
	if (strncmp(fmt, "coordinate", 10) == 0) h->coordinate = 1;
	else if (strncmp(fmt, "array", 5) == 0) h->coordinate = 0;
	else return LISI_ERR(LIS_ERR_FILE_IO, "Not Matrix Market format\n");
	if (strncmp(dtype, "real", 4) != 0) return LISI_ERR(LIS_ERR_FILE_IO, "Not real\n");
	if (strncmp(dstruct, "general", 7) == 0) h->symmetric = 0;
	else if (strncmp(dstruct, "symmetric", 9) == 0) h->symmetric = 1;
	else return LISI_ERR(LIS_ERR_FILE_IO, "Not general or symmetric\n");
	return LIS_SUCCESS;
}

static LIS_INT mm_skip_comments(slurp_t *s, char **b)
{
	char *e;
	do {
		if (!next_line(s, b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	} while ((*b)[0] == '%');
	return LIS_SUCCESS;
}

static LIS_INT mm_size(slurp_t *s, mm_head *h)
{	/* ref lis_input_mm.c:413-458 */
	char *b;
	LISCHK(mm_skip_comments(s, &b));
	int nr = 0, nc = 0, nnz = 0, isb = 0, isx = 0, isbin = 0;
	int got = sscanf(b, "%d %d %d %d %d %d", &nr, &nc, &nnz, &isb, &isx, &isbin);
	if (got < 2) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	if (got == 2) { nnz = nr * nc; isb = isx = isbin = 0; }
	else if (got == 3) { isb = isx = isbin = 0; }
	else if (got == 4) { isx = isbin = 0; }
	else if (got == 5) { isbin = 0; }
	if (nr != nc) return LISI_ERR(LIS_ERR_FILE_IO, "matrix is not square\n");
	h->nr = nr; h->nc = nc; h->nnz = nnz; h->isb = isb; h->isx = isx; h->isbin = isbin;
	return LIS_SUCCESS;
}

/* binary extension records (ref include/lis_io.h:104-115), host byte order unless isbin says otherwise */
typedef struct { int i; int j; double value; } mm_matrec;
typedef struct { int i; double value; } mm_vecrec;
static void bswap4(void *p) { unsigned char *c = (unsigned char *)p, t; t = c[0]; c[0] = c[3]; c[3] = t; t = c[1]; c[1] = c[2]; c[2] = t; }
static void bswap8(void *p) { unsigned char *c = (unsigned char *)p, t; for (int i = 0; i < 4; i++) { t = c[i]; c[i] = c[7 - i]; c[7 - i] = t; } }
static int host_little(void) { int one = 1; return *(char *)&one; }

/* in-file vector section: gn records "idx value" (ref lis_input_mm.c:148-324) */
static LIS_INT mm_read_vec(slurp_t *s, const mm_head *h, LIS_MATRIX A, LIS_VECTOR v)
{
	const int swap = h->isbin && (host_little() != (h->isbin - 1));
	LISCHK(lis_vector_set_size(v, A->n, 0));
	for (LIS_INT i = 0; i < A->gn; i++) {
		long idx; double val;
		if (h->isbin) {
			if (s->pos + sizeof(mm_vecrec) > s->len) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
			mm_vecrec r; memcpy(&r, s->buf + s->pos, sizeof(r)); s->pos += sizeof(r);
			if (swap) { bswap4(&r.i); bswap8(&r.value); }
			idx = r.i; val = r.value;
		} else {
			char *b, *e;
			if (!next_line(s, &b, &e) || !scan_int(&b, &idx) || !scan_double(&b, &val))
				return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
		}
		idx--;
		if (idx >= A->is && idx < A->ie) v->value[idx - A->is] = val;
	}
	lis_amd_vector_host_modified(v);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ the entry lines of a coordinate file, parsed in parallel
 * One line per entry, "row col value" (ref lis_input_mm.c:893-930 reads them with fgets + sscanf("%d %d %lg")).  The body is cut
 * into one piece per host thread at line boundaries, the pieces' line counts give every piece the number of its first entry, and
 * each thread parses its own lines.  Numbers: integers by hand; values by Clinger's exact case -- a decimal mantissa below 2^53
 * times or over a power of ten up to 10^22 is ONE correctly rounded IEEE operation on two exact doubles, i.e. what strtod /
 * sscanf("%lg") return -- and by strtod itself for everything else (long mantissas, large exponents, inf / nan, hex floats).
 * The first malformed line in FILE order is the one reported. */
typedef struct { int kind; LIS_INT entry; long r, c; } mm_text_error;       /* kind 1: malformed / missing line, 2: index outside */

static int scan_double_fast(char **pp, double *out)
{
	static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
	char *p = *pp;
	while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f') p++;
	if (*p == '\n' || *p == '\0') return 0;
	char *q = p;
	int neg = 0;
	if (*q == '+' || *q == '-') { neg = (*q == '-'); q++; }
	unsigned long long m = 0;
	int sig = 0, e10 = 0, any = 0, inexact = 0;
	for (; *q >= '0' && *q <= '9'; q++) {
		any = 1;
		if (sig < 19) { m = m * 10 + (unsigned)(*q - '0'); if (m) sig++; } else { e10++; inexact |= (*q != '0'); }
	}
	if (*q == '.') {
		q++;
		for (; *q >= '0' && *q <= '9'; q++) {
			any = 1;
			if (sig < 19) { m = m * 10 + (unsigned)(*q - '0'); if (m) sig++; e10--; } else inexact |= (*q != '0');
		}
	}
	if (any && (*q == 'e' || *q == 'E')) {
		char *x = q + 1;
		int eneg = 0, ev = 0, edig = 0;
		if (*x == '+' || *x == '-') { eneg = (*x == '-'); x++; }
		for (; *x >= '0' && *x <= '9'; x++) { edig = 1; if (ev < 100000) ev = ev * 10 + (*x - '0'); }
		if (edig) { e10 += eneg ? -ev : ev; q = x; }
	}
	const int alpha = (*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || *q == '.';
	if (any && !alpha && !inexact && m <= (1ULL << 53) && e10 >= -22 && e10 <= 22) {
		double v = (double)m;
		v = e10 < 0 ? v / P10[-e10] : v * P10[e10];
		*out = neg ? -v : v;
		*pp = q;
		return 1;
	}
	char *end;
	const double v = strtod(p, &end);          /* p is not white space: the conversion cannot run into the next line */
	if (end == p) return 0;
	*out = v; *pp = end;
	return 1;
}

static int mm_parse_entries(slurp_t *s, LIS_INT nnz, LIS_INT nr, int *ri, int *ci, double *va, mm_text_error *te)
{
	te->kind = 0; te->entry = nnz; te->r = te->c = 0;
	if (nnz <= 0) return 0;
	char *base = s->buf + s->pos, *end = s->buf + s->len;
	const size_t total = (size_t)(end - base);
	int T = total > ((size_t)4 << 20) ? lisi_host_threads() : 1;
	if (T > 64) T = 64;
	char *cut[65];
	long long first[65];
	cut[0] = base; cut[T] = end;
	for (int t = 1; t < T; t++) {
		char *p = base + total / (size_t)T * (size_t)t;
		char *q = (char *)memchr(p - 1, '\n', (size_t)(end - (p - 1)));
		cut[t] = q ? q + 1 : end;
	}
	long long lines[64];
#pragma omp parallel for num_threads(T) schedule(static, 1)
	for (int t = 0; t < T; t++) {
		long long c = 0;
		char *p = cut[t], *e = cut[t + 1];
		while (p < e) {
			char *q = (char *)memchr(p, '\n', (size_t)(e - p));
			c++;                                   /* (a last line without newline counts too) */
			if (!q) break;
			p = q + 1;
		}
		lines[t] = c;
	}
	first[0] = 0;
	for (int t = 0; t < T; t++) first[t + 1] = first[t] + lines[t];
	mm_text_error errs[64];
	size_t pos_after = s->pos;
#pragma omp parallel for num_threads(T) schedule(static, 1)
	for (int t = 0; t < T; t++) {
		errs[t].kind = 0;
		char *p = cut[t], *e = cut[t + 1];
		for (long long k = first[t]; k < nnz && p < e; k++) {
			char *nl = (char *)memchr(p, '\n', (size_t)(e - p));
			char *le = nl ? nl : e;
			long r, c; double v;
			char *q = p;
			if (!scan_int(&q, &r) || !scan_int(&q, &c) || !scan_double_fast(&q, &v) || q > le) { errs[t].kind = 1; errs[t].entry = (LIS_INT)k; break; }
			if (r < 1 || r > nr || c < 1 || c > nr) { errs[t].kind = 2; errs[t].entry = (LIS_INT)k; errs[t].r = r; errs[t].c = c; break; }
			ri[k] = (int)(r - 1); ci[k] = (int)(c - 1); va[k] = v;
			p = nl ? nl + 1 : e;
			if (k == (long long)nnz - 1) pos_after = (size_t)(p - s->buf);
		}
	}
	for (int t = 0; t < T; t++)
		if (errs[t].kind && errs[t].entry < te->entry) *te = errs[t];
	if (!te->kind && first[T] < nnz) { te->kind = 1; te->entry = (LIS_INT)first[T]; }
	if (te->kind) return 1;
	s->pos = pos_after;
	return 0;
}

/* ------------------------------------------------------------------ coordinate file -> CSR rows [is,ie) */
static LIS_INT mm_read_csr(slurp_t *s, const mm_head *h, LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x)
{
	LISCHK(lis_matrix_set_size(A, 0, h->nr));
	if (A->my_rank == 0) { printf("matrix size = %d x %d (%d nonzero entries)\n\n", h->nr, h->nc, h->nnz); fflush(stdout); }
	const LIS_INT n = A->n, is = A->is, nnz = h->nnz;
	const int swap = h->isbin && (host_little() != (h->isbin - 1));
	LIS_INT err = LIS_SUCCESS;
	int *ri = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
	int *ci = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
	double *va = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
	LIS_INT *ptr = (LIS_INT *)calloc((size_t)n + 1, sizeof(LIS_INT));
	LIS_INT *fill = NULL, *index = NULL;
	LIS_SCALAR *value = NULL;
	if (!ri || !ci || !va || !ptr) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", nnz); goto fail; }

	if (h->isbin) {
		for (LIS_INT k = 0; k < nnz; k++) {
			if (s->pos + sizeof(mm_matrec) > s->len) { err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n"); goto fail; }
			mm_matrec rec; memcpy(&rec, s->buf + s->pos, sizeof(rec)); s->pos += sizeof(rec);
			if (swap) { bswap4(&rec.i); bswap4(&rec.j); bswap8(&rec.value); }
			/* an index outside the matrix would become an x gather index of the HIP kernels (a GPU fault that takes the
			 * context with it); the reference does not check either, but its out-of-bounds read stays in one process */
			if (rec.i < 1 || rec.i > h->nr || rec.j < 1 || rec.j > h->nr) { err = LISI_ERR(LIS_ERR_FILE_IO, "entry %D: index (%D,%D) is outside the matrix\n", k + 1, (LIS_INT)rec.i, (LIS_INT)rec.j); goto fail; }
			ri[k] = rec.i - 1; ci[k] = rec.j - 1; va[k] = rec.value;
		}
	} else {
		mm_text_error te;
		if (mm_parse_entries(s, nnz, h->nr, ri, ci, va, &te)) {
			if (te.kind == 2) err = LISI_ERR(LIS_ERR_FILE_IO, "entry %D: index (%D,%D) is outside the matrix\n", te.entry + 1, (LIS_INT)te.r, (LIS_INT)te.c);
			else err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
			goto fail;
		}
	}
	io_mark("entries parsed");
	/* counting sort by destination row, FILE order kept inside a row (the mirrored entry of a symmetric line before the line's own):
	 * every thread walks all triplets and places the ones whose destination lies in its own range of rows */
	const int T = (nnz > (1 << 16)) ? lisi_host_threads() : 1;
#pragma omp parallel num_threads(T)
	{
		const int t = omp_get_thread_num(), tn = omp_get_num_threads();
		const LIS_INT lo = is + (LIS_INT)((long long)n * t / tn), hi = is + (LIS_INT)((long long)n * (t + 1) / tn);
		for (LIS_INT k = 0; k < nnz; k++) {
			const int r = ri[k], c = ci[k];
			if (h->symmetric && r != c && c >= lo && c < hi) ptr[c - is + 1]++;
			if (r >= lo && r < hi) ptr[r - is + 1]++;
		}
	}
	io_mark("rows counted");
	for (LIS_INT i = 0; i < n; i++) ptr[i + 1] += ptr[i];
	const LIS_INT lnnz = ptr[n];
	index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(lnnz > 0 ? lnnz : 1));
	value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(lnnz > 0 ? lnnz : 1));
	fill = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(n > 0 ? n : 1));
	if (!index || !value || !fill) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", lnnz); goto fail; }
	memcpy(fill, ptr, sizeof(LIS_INT) * (size_t)n);
#pragma omp parallel num_threads(T)
	{
		const int t = omp_get_thread_num(), tn = omp_get_num_threads();
		const LIS_INT lo = is + (LIS_INT)((long long)n * t / tn), hi = is + (LIS_INT)((long long)n * (t + 1) / tn);
		for (LIS_INT k = 0; k < nnz; k++) {
			const int r = ri[k], c = ci[k];
			if (h->symmetric && r != c && c >= lo && c < hi) { const LIS_INT at = fill[c - is]++; index[at] = r; value[at] = va[k]; }
			if (r >= lo && r < hi) { const LIS_INT at = fill[r - is]++; index[at] = c; value[at] = va[k]; }
		}
	}
	io_mark("entries placed");
	free(ri); free(ci); free(va); free(fill);
	ri = ci = NULL; va = NULL; fill = NULL;

	err = lis_matrix_set_csr(lnnz, ptr, index, value, A);
	if (err) goto fail;
	ptr = NULL; index = NULL; value = NULL;          /* adopted */
	{	/* the assemble step is also where a resident / lazily coherent matrix gets its HBM copy and its plan (lisd_mat_eager): timed apart from the reader */
		const double ta = lis_wtime();
		LISCHK(lis_matrix_assemble(A));
		lisg.last_input_assemble_s = lis_wtime() - ta;
	}
	io_mark("assembled");
	if (b != NULL && x != NULL) {
		if (h->isb) LISCHK(mm_read_vec(s, h, A, b));
		if (h->isx) LISCHK(mm_read_vec(s, h, A, x));
	}
	return LIS_SUCCESS;
fail:
	free(ri); free(ci); free(va); free(fill); free(ptr); free(index); free(value);
	return err;
}

/* `array` files: nr*nc values, one per line, column-major (ref lis_input_mm.c:462-697), held by the reference as
 * LIS_MATRIX_DNS and converted to the requested type through CSR keeping value != 0, columns ascending
 * (lis_matrix_dns.c:823-915).  DNS itself is not a served format, so the CSR is built directly. */
static LIS_INT mm_read_dense(slurp_t *s, const mm_head *h, LIS_MATRIX A)
{
	if (h->isbin) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "binary Matrix Market array input is not implemented\n");
	LISCHK(lis_matrix_set_size(A, 0, h->nr));
	if (A->my_rank == 0) { printf("matrix size = %d x %d (%d nonzero entries)\n\n", h->nr, h->nc, h->nr * h->nc); fflush(stdout); }
	const LIS_INT n = A->n, gn = h->nr, is = A->is;
	LIS_INT err = LIS_SUCCESS;
	double *dense = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)gn);
	LIS_INT *ptr = (LIS_INT *)calloc((size_t)n + 1, sizeof(LIS_INT)), *index = NULL;
	LIS_SCALAR *value = NULL;
	if (!dense || !ptr) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", gn); goto fail; }
	for (LIS_INT j = 0; j < gn; j++) {
		for (LIS_INT i = 0; i < gn; i++) {
			char *b, *e; double v;
			if (!next_line(s, &b, &e) || !scan_double(&b, &v)) { err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n"); goto fail; }
			if (i >= is && i < is + n) { dense[(size_t)(i - is) * gn + j] = v; if (v != 0.0) ptr[i - is + 1]++; }
		}
	}
	for (LIS_INT i = 0; i < n; i++) ptr[i + 1] += ptr[i];
	index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(ptr[n] > 0 ? ptr[n] : 1));
	value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(ptr[n] > 0 ? ptr[n] : 1));
	if (!index || !value) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", ptr[n]); goto fail; }
	for (LIS_INT i = 0, k = 0; i < n; i++)
		for (LIS_INT j = 0; j < gn; j++)
			if (dense[(size_t)i * gn + j] != 0.0) { index[k] = j; value[k] = dense[(size_t)i * gn + j]; k++; }
	free(dense); dense = NULL;
	err = lis_matrix_set_csr(ptr[n], ptr, index, value, A);
	if (err) goto fail;
	return lis_matrix_assemble(A);
fail:
	free(dense); free(ptr); free(index); free(value);
	return err;
}

/* ------------------------------------------------------------------ Harwell-Boeing (RUA only, like the reference)
 * lis_input_hb.c:120-463: header lines 1-4 (+5 when right-hand sides are present; they are skipped), fixed-width
 * integer / real fields as described by the Fortran formats of line 4, column pointers / row indices / values =
 * a CSC matrix, converted to CSR unless CSC was requested. */
static int hb_format(const char *field, int size, int *count, int *width)
{	/* "(10I8)" -> 10, 8; "(4E20.12)" / "(1P,4D20.12)"-free forms -> 4, 20 (lis_input_hb.c:100-137) */
	char tmp[64];
	if (size > 63) size = 63;
	memcpy(tmp, field, (size_t)size); tmp[size] = '\0';
	lower(tmp);
	char *p = strchr(tmp, '(');
	*count = 0; *width = 0;
	if (!p) return 1;
	char *sfmt = p + 1, *q = strchr(sfmt, ')');
	if (q) *q = '\0';
	char *k = strchr(sfmt, 'i');
	if (!k) {
		k = strchr(sfmt, 'e');
		if (!k) k = strchr(sfmt, 'd');
		if (!k) return 0;
		char *dot = strchr(sfmt, '.');
		if (dot) *dot = '\0';
	}
	*k = '\0';
	*count = atoi(sfmt);
	*width = atoi(k + 1);
	return 1;
}

static LIS_INT hb_read(slurp_t *s, LIS_MATRIX A)
{
	char *b, *e;
	const LIS_INT want = A->matrix_type;
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");          /* line 1: title, key */
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");          /* line 2: card counts */
	int totcrd = 0, ptrcrd = 0, indcrd = 0, valcrd = 0, rhscrd = 0;
	if (sscanf(b, "%14d%14d%14d%14d%14d", &totcrd, &ptrcrd, &indcrd, &valcrd, &rhscrd) < 4) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");          /* line 3: type, sizes */
	char mtx[64] = "";
	int nrow = 0, ncol = 0, nnzero = 0, neltvl = 0;
	if (sscanf(b, "%63s %d %d %d %d", mtx, &nrow, &ncol, &nnzero, &neltvl) != 5) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	lower(mtx);
	if (mtx[0] != 'r') return LISI_ERR(LIS_ERR_FILE_IO, "Not real\n");
	if (mtx[1] != 'u') return LISI_ERR(LIS_ERR_FILE_IO, "Not unsymmetric\n");
	if (mtx[2] != 'a') return LISI_ERR(LIS_ERR_FILE_IO, "Not assembled\n");
	if (nrow != ncol) return LISI_ERR(LIS_ERR_FILE_IO, "matrix is not square\n");
	if (lisg.rank == 0) { printf("matrix size = %d x %d (%d nonzero entries)\n\n", nrow, ncol, nnzero); fflush(stdout); }
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");          /* line 4: formats */
	char fmt[96];
	memset(fmt, ' ', sizeof(fmt));
	memcpy(fmt, b, (size_t)((e - b) < 95 ? (e - b) : 95));
	int iptr, wptr, iind, wind, ival, wval, irhs, wrhs;
	if (!hb_format(fmt, 16, &iptr, &wptr) || !hb_format(fmt + 16, 16, &iind, &wind) || !hb_format(fmt + 32, 20, &ival, &wval))
		return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	(void)hb_format(fmt + 52, 20, &irhs, &wrhs);
	if (rhscrd != 0 && !next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");   /* line 5 */
	if (lisg.nprocs > 1) return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "Harwell-Boeing input is single-process only\n");
	LISCHK(lis_matrix_set_size(A, 0, nrow));
	const LIS_INT n = A->n;
	LIS_INT *ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * ((size_t)n + 1));
	LIS_INT *index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nnzero > 0 ? nnzero : 1));
	LIS_SCALAR *value = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(nnzero > 0 ? nnzero : 1));
	LIS_INT err = LIS_SUCCESS;
	if (!ptr || !index || !value) { err = LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", nnzero); goto fail; }
	char dat[128];
	for (int pass = 0; pass < 3 && !err; pass++) {
		const int cards = pass == 0 ? ptrcrd : pass == 1 ? indcrd : valcrd;
		const int per = pass == 0 ? iptr : pass == 1 ? iind : ival, wd = pass == 0 ? wptr : pass == 1 ? wind : wval;
		const LIS_INT total = pass == 0 ? n + 1 : nnzero;
		LIS_INT k = 0;
		if (wd <= 0 || wd > 120) { err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n"); break; }
		for (int i = 0; i < cards; i++) {
			if (!next_line(s, &b, &e)) { err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n"); break; }
			char *p = b;
			for (int j = 0; j < per && k < total; j++) {
				int w = 0;
				while (w < wd && p + w < e) { dat[w] = p[w]; w++; }       /* a short last field is what strncpy would copy */
				dat[w] = '\0';
				if (pass == 0) ptr[k] = atoi(dat) - 1;
				else if (pass == 1) index[k] = atoi(dat) - 1;
				else value[k] = atof(dat);
				p += wd;
				if (p > e) p = e;
				k++;
			}
		}
		if (!err && k != total) err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error: section %D holds %D of %D fields\n", (LIS_INT)pass, k, total);
	}
	if (err) goto fail;
	/* column pointers monotone and closed, row indices inside the matrix: they become device addresses */
	if (ptr[0] != 0 || ptr[n] != nnzero) err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error: column pointers do not span the entries\n");
	for (LIS_INT j = 0; j < n && !err; j++) if (ptr[j + 1] < ptr[j]) err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error: column pointers decrease\n");
	for (LIS_INT k = 0; k < nnzero && !err; k++) if (index[k] < 0 || index[k] >= n) err = LISI_ERR(LIS_ERR_FILE_IO, "file i/o error: row index %D is outside the matrix\n", index[k] + 1);
	if (err) goto fail;
	err = lis_matrix_set_csc(nnzero, ptr, index, value, A);
	if (err) goto fail;
	ptr = NULL; index = NULL; value = NULL;
	LISCHK(lis_matrix_assemble(A));
	if (want != LIS_MATRIX_CSC) LISCHK(lisi_matrix_retype(A, LIS_MATRIX_CSR, 0));
	if (want != LIS_MATRIX_CSR && want != LIS_MATRIX_CSC) LISCHK(lisi_matrix_retype(A, want, 0));
	return LIS_SUCCESS;
fail:
	free(ptr); free(index); free(value);
	return err;
}

/* storage type requested with lis_matrix_set_type before the read: convert in place (ref lis_input_mm.c:82-107) */
LIS_INT lisi_matrix_retype(LIS_MATRIX A, LIS_INT want, LIS_INT block)
{
	if (want == A->matrix_type) return LIS_SUCCESS;
	LIS_MATRIX B;
	LISCHK(lis_matrix_duplicate(A, &B));
	if (block > 0) LISCHK(lis_matrix_set_blocksize(B, block, block, NULL, NULL));
	LISCHK(lis_matrix_set_type(B, want));
	LIS_INT err = lis_matrix_convert(A, B);
	if (err) { lis_matrix_destroy(B); return err; }
	lisi_matrix_storage_destroy(A);
	lisi_matrix_copy_header(B, A);
	lisi_unregister(B);
	free(B);
	return LIS_SUCCESS;
}

LIS_INT lis_input(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, char *filename)
{
	if (!lisi_is_registered(A)) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is undefined\n");
	if (A->status != LIS_MATRIX_DECIDING_SIZE && A->status != LIS_MATRIX_NULL)
		return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is already assembled\n");
	if (b != NULL && x != NULL) {
		if (!lisi_is_registered(b) || !lis_vector_is_null(b)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector b is not null\n");
		if (!lisi_is_registered(x) || !lis_vector_is_null(x)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector x is not null\n");
	}
	if (filename == NULL) return LISI_ERR(LIS_ERR_ILL_ARG, "filname is NULL\n");
	slurp_t s;
	const double t_in = lis_wtime();
	lisg.last_input_assemble_s = 0.0; lisg.last_input_s = 0.0;
	LISCHK(slurp_file(filename, &s));
	LIS_INT err;
	if (s.len == 0) { free(s.buf); return LIS_ERR_FILE_IO; }
	if (strncmp(s.buf, MM_BANNER, strlen(MM_BANNER)) != 0) {
		/* anything else is taken for Harwell-Boeing by the reference (lis_input.c:122-125) */
		err = hb_read(&s, A);
		free(s.buf);
		return err;
	}
	const LIS_INT want = A->matrix_type;
	mm_head h;
	memset(&h, 0, sizeof(h));
	err = mm_banner(&s, "matrix", &h);
	if (!err) err = mm_size(&s, &h);
	if (!err && !h.coordinate && want == LIS_MATRIX_DNS)
		err = LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "dense storage (LIS_MATRIX_DNS) is not served by liblis_amd\n");
	if (!err) err = h.coordinate ? mm_read_csr(&s, &h, A, b, x) : mm_read_dense(&s, &h, A);
	free(s.buf);
	if (err) return err;
	err = lisi_matrix_retype(A, want, 0);
	lisg.last_input_s = lis_wtime() - t_in;
	return err;
}
/* seconds of the last lis_input: the whole call, and the part of it spent in lis_matrix_assemble -- which, for a matrix that lives in HBM, is the upload and the plan */
LIS_INT lis_amd_last_input_times(double *total_s, double *assemble_upload_plan_s)
{
	if (total_s) *total_s = lisg.last_input_s;
	if (assemble_upload_plan_s) *assemble_upload_plan_s = lisg.last_input_assemble_s;
	return LIS_SUCCESS;
}

LIS_INT lis_input_matrix(LIS_MATRIX A, char *filename) { return lis_input(A, NULL, NULL, filename); }

/* ------------------------------------------------------------------ vectors in */
static LIS_INT vector_mm(slurp_t *s, LIS_VECTOR v)
{	/* ref lis_input.c:247-389 */
	mm_head h;
	LISCHK(mm_banner(s, "vector", &h));
	if (h.symmetric) return LISI_ERR(LIS_ERR_FILE_IO, "Not general\n");
	char *b;
	LISCHK(mm_skip_comments(s, &b));
	int n;
	if (sscanf(b, "%d", &n) != 1) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	LISCHK(lis_vector_set_size(v, 0, n));
	for (LIS_INT i = 0; i < n; i++) {
		char *e; long idx; double val;
		if (!next_line(s, &b, &e) || !scan_int(&b, &idx) || !scan_double(&b, &val))
			return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
		idx--;
		if (idx >= v->is && idx < v->ie) v->value[idx - v->is] = val;
	}
	return LIS_SUCCESS;
}

static LIS_INT vector_plain(slurp_t *s, LIS_VECTOR v)
{	/* ref lis_input.c:391-438: count the leading numbers of the file, then one number per line */
	LIS_INT n = 0;
	char *p = s->buf;
	for (;;) { double t; if (!scan_double(&p, &t)) break; n++; }
	LISCHK(lis_vector_set_size(v, 0, n));
	for (LIS_INT i = 0; i < n; i++) {
		char *b, *e; double val;
		if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
		if (i >= v->is && i < v->ie) {
			if (!scan_double(&b, &val)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
			v->value[i - v->is] = val;
		}
	}
	return LIS_SUCCESS;
}

static LIS_INT vector_lis_ascii(slurp_t *s, LIS_VECTOR v)
{	/* ref lis_input.c:440-588: "#LIS A vec" / nprocs / "# pe n" / values */
	char *b, *e;
	if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	char banner[16] = "", mode[16] = "", kind[16] = "";
	if (e - b > 10) b[10] = '\0';
	sscanf(b, "%15s %15s %15s", banner, mode, kind);
	if (strncmp(banner, "#LIS", 4) != 0) return LISI_ERR(LIS_ERR_FILE_IO, "not lis file format\n");
	if (mode[0] == 'B' || mode[0] == 'L') return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "binary Lis vector files are not implemented\n");
	if (strncmp(kind, "vec", 3) != 0) return LISI_ERR(LIS_ERR_FILE_IO, "not lis file format\n");
	int np_file;
	if (!next_line(s, &b, &e) || sscanf(b, "%d", &np_file) != 1) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
	if (np_file != lisg.nprocs)
		return LISI_ERR(LIS_ERR_FILE_IO, "The number of PE=(%D) is different (in file PE=%D)\n", (LIS_INT)lisg.nprocs, (LIS_INT)np_file);
	int pe = -1, n = 0;
	do {
		if (!next_line(s, &b, &e)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
		if (b[0] == '#') { char c; if (sscanf(b, "%c %d %d", &c, &pe, &n) != 3) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n"); }
	} while (pe != lisg.rank);
	LISCHK(lis_vector_set_size(v, 0, n));
	char *p = s->buf + s->pos;
	for (LIS_INT i = 0; i < n; i++) {
		double val;
		if (!scan_double(&p, &val)) return LISI_ERR(LIS_ERR_FILE_IO, "file i/o error\n");
		v->value[i] = val;
	}
	return LIS_SUCCESS;
}

LIS_INT lis_input_vector(LIS_VECTOR v, char *filename)
{	/* ref lis_input.c:188-245 */
	if (!lisi_is_registered(v)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector v is undefined\n");
	if (filename == NULL) return LISI_ERR(LIS_ERR_ILL_ARG, "filname is NULL\n");
	slurp_t s;
	LISCHK(slurp_file(filename, &s));
	if (s.len == 0) { free(s.buf); return LIS_ERR_FILE_IO; }
	LIS_INT err;
	if (strncmp(s.buf, MM_BANNER, strlen(MM_BANNER)) == 0) err = vector_mm(&s, v);
	else if (strncmp(s.buf, "#LIS", 4) == 0) err = vector_lis_ascii(&s, v);
	else err = vector_plain(&s, v);
	free(s.buf);
	if (!err) lis_amd_vector_host_modified(v);
	return err;
}

/* ------------------------------------------------------------------ out */
/* ranks take turns in rank order; the token exchange keeps them in step (ref: the pe loops + MPI_Allreduce of
 * lis_output.c:196-246, lis_output_mm.c:85-222) */
static void turn(int nprocs)
{
	if (nprocs > 1) {
		double tok = 0.0, *all = (double *)malloc(sizeof(double) * (size_t)nprocs);
		lisc_allgather_host(&tok, all, sizeof(double));
		free(all);
	}
}

static void put_vec_mm(FILE *f, LIS_VECTOR v, int binary)
{	/* ref lis_output_mm.c:225-324 ("%d %28.20e"; integer vectors "%d %28d") */
	for (LIS_INT i = 0; i < v->n; i++) {
		if (binary) { mm_vecrec r; memset(&r, 0, sizeof(r)); r.i = v->is + i + 1; r.value = v->value[i]; fwrite(&r, sizeof(r), 1, f); }
		else if (v->intvalue) fprintf(f, "%d %28d\n", v->is + i + 1, (int)v->value[i]);
		else fprintf(f, "%d %28.20e\n", v->is + i + 1, (double)v->value[i]);
	}
}

/* CSR (or CSC: same arrays, indices swapped) + optional b, x into one extended Matrix Market file */
static LIS_INT output_mm_to(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_INT format, const char *path)
{	/* ref lis_output_mm.c:327-741 */
	const int binary = (format == LIS_FMT_MMB);
	const int isb = (b && !lis_vector_is_null(b)), isx = (x && !lis_vector_is_null(x));
	if (isb) LISCHK(lisd_vec_to_host(b));
	if (isx) LISCHK(lisd_vec_to_host(x));
	LISCHK(lisp_fill_matrix(A));
	double nnz_local = (double)A->nnz, nnz_total = nnz_local;
	if (A->nprocs > 1) {
		double *all = (double *)malloc(sizeof(double) * (size_t)A->nprocs);
		LISCHK(lisc_allgather_host(&nnz_local, all, sizeof(double)));
		nnz_total = 0.0;
		for (int p = 0; p < A->nprocs; p++) nnz_total += all[p];
		free(all);
	}
	LIS_INT err = LIS_SUCCESS;
	for (int phase = 0; phase < 3; phase++) {          /* matrix entries, then b, then x: each in rank order */
		if (phase == 1 && !isb) continue;
		if (phase == 2 && !isx) continue;
		for (int pe = 0; pe < A->nprocs; pe++) {
			turn(A->nprocs);
			if (pe != A->my_rank) continue;
			FILE *f = fopen(path, (phase == 0 && pe == 0) ? (binary ? "wb" : "w") : (binary ? "ab" : "a"));
			if (!f) { err = LISI_ERR(LIS_ERR_FILE_IO, "cannot open file %s\n", path); continue; }
			if (phase == 0 && pe == 0) {
				fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n");
				if (binary) fprintf(f, "%d %d %d %d %d %d\n", A->gn, A->gn, (int)nnz_total, isb, isx, host_little() + 1);
				else if (!isb && !isx) fprintf(f, "%d %d %d\n", A->gn, A->gn, (int)nnz_total);
				else fprintf(f, "%d %d %d %d %d\n", A->gn, A->gn, (int)nnz_total, isb, isx);
			}
			if (phase == 0) {
				const int csc = (A->matrix_type == LIS_MATRIX_CSC);
				for (LIS_INT i = 0; i < A->n; i++) {
					for (LIS_INT j = A->ptr[i]; j < A->ptr[i + 1]; j++) {
						LIS_INT jj = A->index[j];
						if (A->l2g_map && jj >= A->n) jj = A->l2g_map[jj - A->n]; else jj += A->is;
						const int r = csc ? jj + 1 : A->is + i + 1, c = csc ? A->is + i + 1 : jj + 1;
						if (binary) { mm_matrec rec; rec.i = r; rec.j = c; rec.value = A->value[j]; fwrite(&rec, sizeof(rec), 1, f); }
						else fprintf(f, "%d %d %28.20e\n", r, c, (double)A->value[j]);
					}
				}
			} else put_vec_mm(f, phase == 1 ? b : x, binary);
			fclose(f);
		}
	}
	return err;
}

LIS_INT lis_output(LIS_MATRIX A, LIS_VECTOR b, LIS_VECTOR x, LIS_INT format, char *path)
{	/* ref lis_output.c:63-97 */
	LISCHK(lisi_matrix_check(A, LISI_CHECK_ASSEMBLED));
	if (format != LIS_FMT_MM && format != LIS_FMT_MMB) return LIS_SUCCESS;
	if (MDEV(A)->device_only) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A lives in HBM only (lis_amd_matrix_set_csr_device): nothing to write\n");
	if (A->matrix_type == LIS_MATRIX_CSR) return output_mm_to(A, b, x, format, path);
	LIS_MATRIX B;
	LISCHK(lis_matrix_duplicate(A, &B));
	LISCHK(lis_matrix_set_type(B, LIS_MATRIX_CSR));
	LIS_INT err = lis_matrix_convert(A, B);
	if (!err) err = output_mm_to(B, b, x, format, path);
	lis_matrix_destroy(B);
	return err;
}

LIS_INT lis_output_matrix(LIS_MATRIX A, LIS_INT format, char *path) { return lis_output(A, NULL, NULL, format, path); }

LIS_INT lis_output_vector(LIS_VECTOR v, LIS_INT format, char *filename)
{	/* ref lis_output.c:146-583: PLAIN one "%28.20e" per line; MM "vector coordinate real general" + gn + "i v";
	 * LIS "#LIS A vec" / nprocs / "# pe n" / three "%28.20e " per line */
	if (!lisi_is_registered(v)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector v is undefined\n");
	if (lis_vector_is_null(v)) return LISI_ERR(LIS_ERR_ILL_ARG, "vector v is null\n");
	if (format != LIS_FMT_PLAIN && format != LIS_FMT_MM && format != LIS_FMT_LIS) return LISI_ERR(LIS_ERR_ILL_ARG, "ill format option\n");
	LISCHK(lisd_vec_to_host(v));
	LIS_INT err = LIS_SUCCESS;
	for (LIS_INT pe = 0; pe < v->nprocs; pe++) {
		turn(v->nprocs);
		if (pe != v->my_rank) continue;
		FILE *f = fopen(filename, pe == 0 ? "w" : "a");
		if (!f) { err = LISI_ERR(LIS_ERR_FILE_IO, "cannot open file %s\n", filename); continue; }
		if (pe == 0 && format == LIS_FMT_MM)
			fprintf(f, "%%%%MatrixMarket vector coordinate %s general\n%d\n", v->intvalue ? "integer" : "real", v->gn);
		if (pe == 0 && format == LIS_FMT_LIS) fprintf(f, "#LIS A vec\n%d\n", v->nprocs);
		if (format == LIS_FMT_MM) put_vec_mm(f, v, 0);
		else if (format == LIS_FMT_PLAIN) {
			for (LIS_INT i = 0; i < v->n; i++) {
				if (v->intvalue) fprintf(f, "%28d\n", (int)v->value[i]);
				else fprintf(f, "%28.20e\n", (double)v->value[i]);
			}
		} else {
			fprintf(f, "# %d %d\n", (int)pe, v->n);
			for (LIS_INT i = 0; i < v->n; i++) {
				if (v->intvalue) fprintf(f, "%28d ", (int)v->value[i]);
				else fprintf(f, "%28.20e ", (double)v->value[i]);
				if ((i + 1) % 3 == 0) fprintf(f, "\n");
			}
			if (v->n % 3 != 0) fprintf(f, "\n");
		}
		fclose(f);
	}
	return err;
}
