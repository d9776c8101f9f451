/*
 * lis_system.c -- process-level services of the Lis API: init/finalize, allocation, errors, timers.
 *
 * Mirrors the caller-visible behaviour of the reference's src/system (lis_init.c:121-247,
 * lis_memory.c, lis_error.c:161-191, lis_time.c); none of it is on the hot path.
 */
#define _GNU_SOURCE
#include <stdarg.h>
#include <stdio.h>
#include <sys/time.h>
#include "lis_internal.h"

lisi_globals lisg = {0};
int lisi_cmd_argc = 0;
char **lisi_cmd_argv = NULL;

/* ------------------------------------------------------------------ registry of live handles */
typedef struct { void *obj; int kind; } reg_entry;
static reg_entry *reg_tab = NULL;
static size_t reg_cap = 0, reg_live = 0, reg_used = 0;   /* open addressing, tombstone = (void*)1; used = live + tombstones */

static size_t reg_hash(const void *p, size_t cap) { return (((size_t)p) >> 4) * 0x9E3779B97F4A7C15ULL % cap; }

/* rebuild the table from its live entries: twice the size when they fill half of it, the same size when it is the
 * tombstones of a long create/destroy history that do (the table follows the LIVE count, not the total ever made) */
static void reg_rehash(void)
{
	size_t ncap = reg_cap ? reg_cap : 256;
	while ((reg_live + 1) * 2 > ncap) ncap *= 2;
	reg_entry *nt = (reg_entry *)calloc(ncap, sizeof(reg_entry));
	for (size_t i = 0; i < reg_cap; i++) {
		if (reg_tab[i].obj && reg_tab[i].obj != (void *)1) {
			size_t h = reg_hash(reg_tab[i].obj, ncap);
			while (nt[h].obj) h = (h + 1) % ncap;
			nt[h] = reg_tab[i];
		}
	}
	free(reg_tab);
	reg_tab = nt; reg_cap = ncap; reg_used = reg_live;
}

void lisi_register(void *obj, int kind)
{
	if ((reg_used + 1) * 2 > reg_cap) reg_rehash();
	size_t h = reg_hash(obj, reg_cap);
	while (reg_tab[h].obj && reg_tab[h].obj != (void *)1) h = (h + 1) % reg_cap;
	if (!reg_tab[h].obj) reg_used++;           /* a reused tombstone was counted already */
	reg_tab[h].obj = obj; reg_tab[h].kind = kind;
	reg_live++;
}

static reg_entry *reg_find(void *obj)
{
	if (!reg_cap || !obj || obj == (void *)1) return NULL;
	size_t h = reg_hash(obj, reg_cap);
	for (size_t probes = 0; probes < reg_cap && reg_tab[h].obj; probes++, h = (h + 1) % reg_cap)
		if (reg_tab[h].obj == obj) return &reg_tab[h];
	return NULL;
}

void lisi_unregister(void *obj) { reg_entry *e = reg_find(obj); if (e) { e->obj = (void *)1; reg_live--; } }
size_t lisi_registry_slots(void) { return reg_cap; }          /* tests: the table must not grow with the create/destroy count */
int  lisi_is_registered(void *obj) { return reg_find(obj) != NULL; }

/* ------------------------------------------------------------------ host threads for the one-off conversions (lis_convert.c)
 * A container sees every core of the machine but may only run a quota of them: a team of 128 threads on a 16-core quota is
 * throttled to a crawl (measured: csr2dia at 256^3 11 s instead of 0.5 s).  Affinity mask, capped by cpu.max, capped at 32;
 * LIS_AMD_HOST_THREADS overrides. */
#include <sched.h>
int lisi_host_threads(void)
{
	static int cached = 0;
	if (cached) return cached;
	int n = 1;
	const char *env = getenv("LIS_AMD_HOST_THREADS");
	if (env && atoi(env) > 0) n = atoi(env);
	else {
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
		FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
		if (f) {
			char quota[64]; long period = 0;
			if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
				const long q = atol(quota) / period;
				if (q >= 1 && q < n) n = (int)q;
			}
			fclose(f);
		}
		if (n > 32) n = 32;
	}
	if (n < 1) n = 1;
	cached = n;
	return n;
}

/* ------------------------------------------------------------------ allocation (ref:1037-1042)
 * The reference keeps a list of every block so that lis_free_all can sweep at finalize; callers may
 * also hand malloc'ed arrays to lis_matrix_set_* and have lis_free fall back to free().  Plain
 * malloc/free gives the same observable behaviour. */
void *lis_malloc(size_t size, char *tag) { (void)tag; return malloc(size ? size : 1); }
void *lis_calloc(size_t size, char *tag) { (void)tag; return calloc(size ? size : 1, 1); }
/* (arrays of lis_matrix_malloc_<fmt> live on pages of the library's, lis_pages.c: lis_free / lis_realloc know them, free() does not) */
void *lis_realloc(void *p, size_t size)
{
	const size_t have = lisp_array_bytes(p);
	if (!have) return realloc(p, size ? size : 1);
	void *q = malloc(size ? size : 1);
	if (q) { memcpy(q, p, have < size ? have : size); (void)lisp_free_array(p); }
	return q;
}
void  lis_free(void *p) { if (!lisp_free_array(p)) free(p); }
void  lis_free2(LIS_INT n, ...)
{
	va_list ap;
	va_start(ap, n);
	for (LIS_INT i = 0; i < n; i++) { void *p = va_arg(ap, void *); if (p) lis_free(p); }
	va_end(ap);
}
LIS_INT lis_is_malloc(void *p) { return lisi_is_registered(p) ? LIS_TRUE : LIS_FALSE; }

/* ------------------------------------------------------------------ printing and errors */
static const char *err_name(LIS_INT code)
{
	static const char *names[] = {"ILL_ARG", "BREAKDOWN", "OUT_OF_MEMORY", "MAXITER", "NOT_IMPLEMENTED", "FILE_IO_ERROR"};
	return (code >= 1 && code <= 6) ? names[code - 1] : "UNKNOWN";
}

/* the reference's format strings use %D for LIS_INT (lis_error.c:124) */
static void vprint_lis(FILE *f, const char *fmt, va_list ap)
{
	char buf[2048];
	size_t w = 0;
	for (const char *p = fmt; *p && w + 2 < sizeof(buf); p++) {
		if (p[0] == '%' && p[1] == 'D') { buf[w++] = '%'; buf[w++] = 'd'; p++; }
		else buf[w++] = *p;
	}
	buf[w] = 0;
	vfprintf(f, buf, ap);
}

LIS_INT lis_printf(LIS_Comm comm, const char *mess, ...)
{
	(void)comm;
	if (lisg.rank != 0) return LIS_SUCCESS;
	va_list ap;
	va_start(ap, mess);
	vprint_lis(stdout, mess, ap);
	va_end(ap);
	return LIS_SUCCESS;
}

LIS_INT lisi_error(const char *file, const char *func, int line, LIS_INT code, const char *fmt, ...)
{
	if (lisg.rank != 0) return LIS_SUCCESS;
	const char *base = strrchr(file, '/');
	printf("%s(%d) : %s : error %s :", base ? base + 1 : file, line, func, err_name(code));
	if (fmt) { va_list ap; va_start(ap, fmt); vprint_lis(stdout, fmt, ap); va_end(ap); }
	fflush(stdout);
	return LIS_SUCCESS;
}

LIS_INT lisi_hip_error(const char *file, const char *func, int line, int hipcode)
{
	const char *base = strrchr(file, '/');
	if (hipcode == LISHIP_ERR_TIMEOUT) {
		/* the watchdog of a multi-rank job (lis_comm.c): a collective or a halo exchange this rank queued never completed.  Nothing useful can follow -- the
		 * communicator is wedged and every later call would wait again -- so the job dies here, loudly, instead of hanging until somebody notices */
		fprintf(stderr, "%s(%d) : %s : rank %d of %d: the stream did not drain within the communication time limit (LIS_AMD_COMM_TIMEOUT seconds): a peer of the RCCL "
		        "communicator is missing or stuck in another collective -- aborting\n", base ? base + 1 : file, line, func, (int)lisg.rank, (int)(lisg.nprocs ? lisg.nprocs : 1));
		fflush(stderr);
		abort();
	}
	fprintf(stderr, "%s(%d) : %s : HIP error %d (%s) -- liblis_amd has no CPU fallback\n",
	        base ? base + 1 : file, line, func, hipcode, liship_error_string(hipcode));
	return hipcode == 2 /* hipErrorOutOfMemory */ ? LIS_ERR_OUT_OF_MEMORY : LIS_AMD_ERR_DEVICE;
}

void CHKERR(LIS_INT err)
{
	if (err) { lis_finalize(); exit((int)err); }
}

double lis_wtime(void)
{	/* Drivers bracket what they time with lis_wtime() (test/spmvtest*.c, test/test*.c).  In resident mode the calls in
	 * between only QUEUE work, so the clock is read after the queue has drained: the interval is the work's. */
	if (lisg.device_ready && lisg.stream) (void)liship_stream_synchronize(lisg.stream);
	struct timeval tv;
	gettimeofday(&tv, NULL);
	return (double)tv.tv_sec + (double)tv.tv_usec * 1.0e-6;
}

/* ------------------------------------------------------------------ argument tokens */
int lisi_tokenize(const char *text, char ***tokens)
{
	int n = 0, cap = 16;
	char **t = (char **)malloc(sizeof(char *) * cap);
	const char *p = text;
	while (p && *p) {
		while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
		if (!*p) break;
		const char *q = p;
		while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;
		if (n == cap) { cap *= 2; t = (char **)realloc(t, sizeof(char *) * cap); }
		t[n] = (char *)malloc((size_t)(q - p) + 1);
		memcpy(t[n], p, (size_t)(q - p));
		t[n][q - p] = 0;
		n++;
		p = q;
	}
	*tokens = t;
	return n;
}

void lisi_tokens_free(char **tokens, int n)
{
	for (int i = 0; i < n; i++) free(tokens[i]);
	free(tokens);
}

/* ------------------------------------------------------------------ init / finalize */
LIS_INT lis_initialize(int *argc, char **argv[])
{
	/* the command line is kept for lis_solver_set_optionC (ref lis_init.c:148, lis_solver.c:1095).
	 * -omp_num_threads is accepted and ignored: the parallelism here is the GPU's. */
	if (lisi_cmd_argv) { lisi_tokens_free(lisi_cmd_argv, lisi_cmd_argc); lisi_cmd_argv = NULL; lisi_cmd_argc = 0; }
	if (argc && argv && *argc > 0 && *argv) {
		lisi_cmd_argc = *argc;
		lisi_cmd_argv = (char **)malloc(sizeof(char *) * (size_t)*argc);
		for (int i = 0; i < *argc; i++) {
			const char *s = (*argv)[i] ? (*argv)[i] : "";
			lisi_cmd_argv[i] = (char *)malloc(strlen(s) + 1);
			strcpy(lisi_cmd_argv[i], s);
		}
		for (int i = 1; i < *argc; i++) {
			if (strncmp(lisi_cmd_argv[i], "-ver", 4) == 0) { lis_printf(LIS_COMM_WORLD, "Lis Version %s (lis_amd, gfx950)\n", LIS_VERSION); CHKERR(1); }
			if (strncmp(lisi_cmd_argv[i], "-help", 5) == 0) CHKERR(1);
		}
	}
	if (!lisg.initialized) {
		lisg.initialized = 1;
		if (lisg.nprocs == 0) { lisg.nprocs = 1; lisg.rank = 0; }
		const char *r = getenv("LIS_AMD_RESIDENCY");
		if (r && (strcmp(r, "resident") == 0 || strcmp(r, "1") == 0)) lisg.residency = LIS_AMD_RESIDENT;
		r = getenv("LIS_AMD_COHERENCE");              /* eager: COHERENT copies on every call instead of following page faults (lis_pages.c) */
		lisg.eager_coherence = (r && strcmp(r, "eager") == 0);
		r = getenv("LIS_AMD_NO_FUSION");
		lisg.no_fusion = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_OVERLAP");
		lisg.no_overlap = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_DIRECT_HALO");
		lisg.no_direct_halo = (r && r[0] == '1');
		r = getenv("LIS_AMD_MATRIX_CHECK");
		lisg.matrix_check = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_INDEX_CODES");
		lisg.no_index_codes = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_ROW_PATTERNS");
		lisg.no_row_patterns = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_VALUE_RECORDS");
		lisg.no_value_records = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_DEVICE_CONVERT");
		lisg.no_device_convert = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_ROW_FORM");
		lisg.no_row_form = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_UNIFORM_JACOBI");
		lisg.no_uniform_jacobi = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_LOCAL_COLUMNS");
		lisg.no_local_columns = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_REORDER");
		lisg.no_reorder = (r && r[0] == '1');
		r = getenv("LIS_AMD_PLAIN_MALLOC");            /* lis_matrix_malloc_<fmt> hands out malloc memory (a program that free()s the arrays or read()s into them) */
		lisg.plain_malloc = (r && r[0] == '1');
		r = getenv("LIS_AMD_REORDER_AFTER");           /* products a plan serves before a lis_solve builds its renumbered form (0: at plan time, the round-5 behaviour) */
		lisg.reorder_after = r ? atoll(r) : 4096;
		if (lisg.reorder_after < 0) lisg.reorder_after = 0;
		r = getenv("LIS_AMD_REORDER_PRODUCTS");        /* single products of renumbered long-row plans take the renumbered form too (gather of x, scattered store of y): opt-in */
		if (r && r[0] == '1') (void)liship_spmv_csr_set_reorder(2);
		r = getenv("LIS_AMD_NO_MARCHING");
		lisg.no_marching = (r && r[0] == '1');
		r = getenv("LIS_AMD_NO_TEAM_KERNELS");
		lisg.no_team_kernels = (r && r[0] == '1');
		r = getenv("LIS_AMD_ROW_BLOCK_DOTS");         /* the fused dots of the dominant-pattern product as the row blocks' partial sums: every form's bits, slower */
		lisg.row_block_dots = (r && r[0] == '1');
		r = getenv("LIS_AMD_REFERENCE_LAYOUT");      /* products stream the reference's own arrays: no codes, patterns, value records, local columns, renumbering, row forms */
		lisg.reference_layout = (r && r[0] == '1');
		r = getenv("LIS_AMD_LONG_ROW_CHAIN");         /* the part of a row beyond the LDS stage as ONE left-to-right chain: the reference's bits for hub rows, slow on them */
		lisg.long_row_chain = (r && r[0] == '1');
		r = getenv("LIS_AMD_LONG_ROW_TREE");          /* (rounds 4-5 spelling: the tree was opt-in then; =0 still selects the chain) */
		if (r && r[0] == '0') lisg.long_row_chain = 1;
		r = getenv("LIS_AMD_REFERENCE_REDUCTIONS");   /* T: every sum in the reference's order for OMP_NUM_THREADS = T (parity mode, slow) */
		lisg.ref_reductions = (r && atoi(r) > 0) ? atoi(r) : 0;
		if (lisg.device_ready) (void)liship_set_reference_reductions(lisg.ref_reductions);      /* (a second lis_initialize in one process) */
		r = getenv("LIS_AMD_GRAPHS");
		lisg.graphs = (r && r[0] == '1');
		r = getenv("LIS_AMD_HOST_SCALARS");
		lisg.host_scalars = (r && r[0] == '1');
		/* RESIDENT, and COHERENT by page protection (which runs at resident speed, so it starts like RESIDENT): the runtime comes up here (quietly: a box
		 * without a GPU still serves the host-side API), and matrices are uploaded where they are made (lisd_mat_eager).  BEHIND the switches above: the
		 * device's start-up applies some of them (LIS_AMD_NO_TEAM_KERNELS, LIS_AMD_ROW_BLOCK_DOTS, LIS_AMD_LONG_ROW_CHAIN) */
		if (lisg.residency == LIS_AMD_RESIDENT || (lisg.residency == LIS_AMD_COHERENT && !lisg.eager_coherence)) (void)lisd_init_quiet();
	}
	return LIS_SUCCESS;
}

LIS_INT lis_finalize(void)
{
	if (lisi_cmd_argv) { lisi_tokens_free(lisi_cmd_argv, lisi_cmd_argc); lisi_cmd_argv = NULL; lisi_cmd_argc = 0; }
	lisg.initialized = 0;
	(void)lis_amd_trim();
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ sorting helpers
 * ascending sort of i1[is..ie] carrying d1 (ref src/system/lis_sort.c:90-118).  Drivers call it on rows
 * with distinct columns, where any correct sort gives the reference's order. */
void lis_sort_id(LIS_INT is, LIS_INT ie, LIS_INT *i1, LIS_SCALAR *d1)
{
	if (ie > is) lisi_sort_row(is, ie + 1, i1, d1);
}

void lisi_sort_row(LIS_INT lo, LIS_INT hi, LIS_INT *idx, LIS_SCALAR *val)
{
	/* [lo,hi): insertion sort for short rows, heap sort otherwise (no recursion, O(n log n) worst case) */
	LIS_INT len = hi - lo;
	if (len < 2) return;
	LIS_INT *k = idx + lo;
	LIS_SCALAR *v = val + lo;
	if (len <= 24) {
		for (LIS_INT a = 1; a < len; a++) {
			LIS_INT ck = k[a]; LIS_SCALAR cv = v[a]; LIS_INT b = a - 1;
			while (b >= 0 && k[b] > ck) { k[b + 1] = k[b]; v[b + 1] = v[b]; b--; }
			k[b + 1] = ck; v[b + 1] = cv;
		}
		return;
	}
	for (LIS_INT start = len / 2 - 1, end = len;;) {
		LIS_INT root;
		if (start >= 0) root = start--;
		else {
			if (--end <= 0) break;
			LIS_INT tk = k[0]; k[0] = k[end]; k[end] = tk;
			LIS_SCALAR tv = v[0]; v[0] = v[end]; v[end] = tv;
			root = 0;
		}
		for (;;) {
			LIS_INT child = 2 * root + 1;
			if (child >= end) break;
			if (child + 1 < end && k[child + 1] > k[child]) child++;
			if (k[root] >= k[child]) break;
			LIS_INT tk = k[root]; k[root] = k[child]; k[child] = tk;
			LIS_SCALAR tv = v[root]; v[root] = v[child]; v[child] = tv;
			root = child;
		}
	}
}
