/*
 * lis_device.c -- HBM copies of Lis objects and the dispatch of the hot path onto the HIP kernels.
 *
 * Data layout in HBM (all f64 / i32):
 *   vectors   one buffer of np + pad (+ slack) doubles; owned entries [0,n), ghost entries [n,np)
 *   CSR       ptr[n+1], index[nnz], value[nnz] + the merge-path row split (liship_csr_plan_t)
 *   CSC       kept as the column-ordered transpose, i.e. CSR whose rows list their entries by ascending
 *             column: the reference's serial CSC loop (src/matvec/lis_matvec_csc.c:128-144) adds the
 *             terms of output row i in exactly that order
 *   ELL/DIA   column-major [maxnzr|nnd][n];  JAD  perm/ptr/index/value (one chunk);  BSR  bptr/bindex/value
 */
#include <stdio.h>
#include "lis_internal.h"

/* ------------------------------------------------------------------ runtime */
/* resident mode pays the one-off costs (runtime start-up, matrix upload, row split) where the reference pays its
 * own (lis_initialize, lis_matrix_assemble / _convert), so a driver that times a loop of lis_matvec times products */
LIS_INT lisd_init_quiet(void)
{
	int count = 0;
	if (lisg.device_ready) return LIS_SUCCESS;
	if (liship_device_count(&count) != 0 || count < 1) return LIS_ERR_NOT_IMPLEMENTED;
	return lisd_init();
}

/* the HBM copy of a matrix is built where the matrix is made (assemble / convert) once this process has used the GPU at all: the upload and
 * the plan belong to the conversion, not to the first product a driver times.  RESIDENT, and COHERENT by page protection (whose products
 * then run at resident speed from the first one on); eager COHERENT keeps building it on first use. */
void lisd_mat_eager(LIS_MATRIX A)
{
	if ((lisg.residency == LIS_AMD_RESIDENT || lisp_lazy()) && lisg.device_ready && A->status >= LIS_MATRIX_CSR) (void)lisd_mat_ready(A);
}

static LIS_INT stage_ready(void);
LIS_INT lisd_init(void)
{
	if (lisg.device_ready) return LIS_SUCCESS;
	int count = 0;
	int rc = liship_device_count(&count);
	if (rc != 0 || count < 1) {
		fprintf(stderr, "liblis_amd: no HIP device available (%s) -- this library has no CPU fallback\n",
		        rc ? liship_error_string(rc) : "device count is 0");
		return LIS_ERR_NOT_IMPLEMENTED;
	}
	if (lisg.comm_kind != 1) {           /* the RCCL bootstrap already chose the device */
		/* LIS_AMD_DEVICE, else the launcher's LOCAL_RANK (torchrun, mpirun wrappers: one process per GPU), else device 0 */
		const char *env = getenv("LIS_AMD_DEVICE");
		if (env) lisg.device = atoi(env);
		else { env = getenv("LOCAL_RANK"); lisg.device = env ? atoi(env) % count : 0; }
		if (lisg.device < 0 || lisg.device >= count) lisg.device = 0;
	}
	HIPCHK(liship_set_device(lisg.device));
	HIPCHK(liship_stream_create(&lisg.stream));
	HIPCHK(lisd_malloc(&lisg.reduce_work, liship_reduce_work_bytes()));
	HIPCHK(lisd_malloc((void **)&lisg.reduce_out, 4 * sizeof(double)));
	HIPCHK(liship_malloc_host((void **)&lisg.host_out, 4 * 64 * sizeof(double)));
	HIPCHK(liship_spmv_csr_set_long_row_tree((lisg.long_row_chain || lisg.ref_reductions) ? 0 : 1));     /* parity modes take the chain */
	if (lisg.reference_layout) LISCHK(lis_amd_set_reference_layout(1));
	if (lisg.no_team_kernels) { HIPCHK(liship_spmv_csr_set_team(0)); HIPCHK(liship_spmv_bsr_set_team(0)); }
	if (lisg.no_marching) HIPCHK(liship_spmv_csr_set_dom_march(0));
	{ const char *e = getenv("LIS_AMD_NO_LOCAL_SHORT_ROWS"); lisg.no_local_short_rows = (e && e[0] == '1'); }      /* A/B: short-row plans never try block-local columns (csr_plan_impl) */
	if (lisg.row_block_dots) HIPCHK(liship_spmv_csr_set_row_block_dots(1));
	if (lisg.ref_reductions) HIPCHK(liship_set_reference_reductions(lisg.ref_reductions));
	lisg.device_ready = 1;
	if (lisp_lazy()) (void)stage_ready();      /* the pinned staging buffers of the page-protected vectors: here, not inside a caller's first product */
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ work-vector pool
 * A Krylov solve needs 4 ... restart+3 vectors of the matrix size; hipMalloc/hipFree of tens of GiB cost more
 * than the iterations of a short solve (GMRES(30) at 512^3: 33 GiB, ~1 s).  Buffers released by a solve are
 * kept and handed to the next one that asks for the same size; lis_amd_trim() / lis_finalize() / an allocation
 * failure anywhere give them back to the driver. */
#define POOL_SLOTS 160
static struct { void *p; size_t bytes; } pool[POOL_SLOTS];

/* every HBM allocation of the host layer: when the driver is out of memory the pooled work vectors (up to 33 GiB
 * after a GMRES(30) solve at 512^3) go back to it and the request is tried once more.  Returns the HIP code. */
int lisd_malloc(void **out, size_t bytes)
{
	int rc = liship_malloc(out, bytes);
	if (rc && lis_amd_trim_count() > 0) rc = liship_malloc(out, bytes);
	return rc;
}

LIS_INT lisd_pool_get(size_t bytes, void **out)
{
	for (int i = 0; i < POOL_SLOTS; i++)
		if (pool[i].p && pool[i].bytes == bytes) { *out = pool[i].p; pool[i].p = NULL; return LIS_SUCCESS; }
	HIPCHK(lisd_malloc(out, bytes));
	return LIS_SUCCESS;
}

void lisd_pool_put(void *p, size_t bytes)
{
	if (!p) return;
	for (int i = 0; i < POOL_SLOTS; i++)
		if (!pool[i].p) { pool[i].p = p; pool[i].bytes = bytes; return; }
	(void)liship_free(p);
}

/* gives the pooled buffers back; returns how many there were */
int lis_amd_trim_count(void)
{
	int freed = 0;
	for (int i = 0; i < POOL_SLOTS; i++)
		if (pool[i].p) { (void)liship_free(pool[i].p); pool[i].p = NULL; pool[i].bytes = 0; freed++; }
	return freed;
}
static void renum_cache_drop(void);
LIS_INT lis_amd_trim(void) { (void)lis_amd_trim_count(); renum_cache_drop(); return LIS_SUCCESS; }

void *lis_amd_stream(void) { return lisd_init() == LIS_SUCCESS ? lisg.stream : NULL; }

LIS_INT lis_amd_synchronize(void)
{
	LISCHK(lisd_init());
	HIPCHK(liship_device_synchronize());
	return LIS_SUCCESS;
}

LIS_INT lis_amd_set_residency(LIS_INT mode)
{
	if (mode != LIS_AMD_COHERENT && mode != LIS_AMD_RESIDENT) return LISI_ERR(LIS_ERR_ILL_ARG, "unknown residency mode %D\n", mode);
	lisg.residency = mode;
	if (mode == LIS_AMD_RESIDENT) (void)lisd_init_quiet();   /* runtime start-up now, not inside the first product */
	return LIS_SUCCESS;
}
LIS_INT lis_amd_get_residency(void) { return lisg.residency; }

LIS_INT lis_amd_last_solve_uniform_jacobi(void) { return lisg.last_uniform_jacobi; }

LIS_INT lis_amd_set_row_form(LIS_INT on) { lisg.no_row_form = on ? 0 : 1; return LIS_SUCCESS; }
/* The REFERENCE LAYOUT mode (env LIS_AMD_REFERENCE_LAYOUT=1): every product streams the reference's own arrays -- CSR 4 B index[] + 8 B value[] per non-zero + ptr[]
 * (lis_matvec_csr.c:97-109 on SURVEY 8d's 12 B per non-zero + 20 B per row), ELL / DIA / BSR their native arrays -- and nothing the plan could derive from them: no
 * one-byte column codes, row patterns, value records, block-local columns, renumbering or row forms.  Same bits either way; this is the form the CSR roofline target is
 * quoted on and what bench.py's headline runs.  Applies to plans already built (the kernels' own switches) and to matrices uploaded from now on. */
static struct { int saved, codes, patterns, values, rowform, local, reorder; } ref_layout;
LIS_INT lis_amd_set_reference_layout(LIS_INT on)
{
	if (on && !ref_layout.saved) {
		ref_layout.saved = 1;
		ref_layout.codes = lisg.no_index_codes; ref_layout.patterns = lisg.no_row_patterns; ref_layout.values = lisg.no_value_records;
		ref_layout.rowform = lisg.no_row_form; ref_layout.local = lisg.no_local_columns; ref_layout.reorder = lisg.no_reorder;
		lisg.no_index_codes = lisg.no_row_patterns = lisg.no_value_records = lisg.no_row_form = lisg.no_local_columns = lisg.no_reorder = 1;
	} else if (!on && ref_layout.saved) {
		ref_layout.saved = 0;
		lisg.no_index_codes = ref_layout.codes; lisg.no_row_patterns = ref_layout.patterns; lisg.no_value_records = ref_layout.values;
		lisg.no_row_form = ref_layout.rowform; lisg.no_local_columns = ref_layout.local; lisg.no_reorder = ref_layout.reorder;
	}
	lisg.reference_layout = on ? 1 : 0;
	HIPCHK(liship_spmv_csr_set_index_codes(!lisg.no_index_codes)); HIPCHK(liship_spmv_csr_set_row_patterns(!lisg.no_row_patterns));
	HIPCHK(liship_spmv_csr_set_row_values(!lisg.no_value_records)); HIPCHK(liship_spmv_csr_set_local_columns(!lisg.no_local_columns));
	HIPCHK(liship_spmv_csr_set_reorder(lisg.no_reorder ? 0 : 1));
	return LIS_SUCCESS;
}
LIS_INT lis_amd_get_reference_layout(void) { return lisg.reference_layout; }
LIS_INT lis_amd_set_reference_reductions(LIS_INT T)
{
	if (T < 0 || liship_set_reference_reductions((int)T) != 0) return LISI_ERR(LIS_ERR_ILL_ARG, "reference-order reductions: T(=%D) out of range\n", T);
	lisg.ref_reductions = (int)T;
	/* the parity mode wants the reference's bits everywhere: hub rows go back to the left-to-right chain while it is on */
	if (lisg.device_ready) HIPCHK(liship_spmv_csr_set_long_row_tree((lisg.long_row_chain || T > 0) ? 0 : 1));
	return LIS_SUCCESS;
}
LIS_INT lis_amd_get_reference_reductions(void) { return lisg.ref_reductions; }
LIS_INT lis_amd_set_graphs(LIS_INT on) { lisg.graphs = (on != 0); return LIS_SUCCESS; }
LIS_INT lis_amd_last_solve_graph_replays(void) { return lisg.last_graph_replays; }
LIS_INT lis_amd_last_solve_renumbered(void) { return lisg.last_renumbered; }
LIS_INT lis_amd_set_loop_mode(LIS_INT mode)
{
	if (mode < LIS_AMD_LOOP_DEVICE || mode > LIS_AMD_LOOP_UNFUSED) return LISI_ERR(LIS_ERR_ILL_ARG, "unknown loop mode %D\n", mode);
	lisg.host_scalars = (mode == LIS_AMD_LOOP_HOST);
	lisg.no_fusion = (mode == LIS_AMD_LOOP_UNFUSED);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ vectors */
static size_t vec_len(LIS_VECTOR v) { return (size_t)(v->np + v->pad); }

LIS_INT lisd_vec_reserve(LIS_VECTOR v, size_t doubles)
{
	lisd_vec *d = VDEV(v);
	LISCHK(lisd_init());
	if (d->d && d->cap >= doubles) return LIS_SUCCESS;
	const size_t cap = doubles + 16;                      /* slack: BSR kernels read/write whole blocks */
	void *nd = NULL;
	HIPCHK(lisd_malloc(&nd, cap * sizeof(double)));
	HIPCHK(liship_memset(nd, 0, cap * sizeof(double), lisg.stream));
	if (d->d) {
		if (d->dev_valid) HIPCHK(liship_memcpy_d2d(nd, d->d, d->cap * sizeof(double), lisg.stream));
		HIPCHK(liship_stream_synchronize(lisg.stream));
		HIPCHK(liship_free(d->d));
	}
	d->d = (double *)nd;
	d->cap = cap;
	return LIS_SUCCESS;
}

/* Copies between page-protected host arrays and HBM go through two pinned staging buffers.  Handing the vector's own pages to the runtime
 * would make it register them with the driver (a pageable copy of this size pins its source), and every later mprotect of registered pages
 * is an MMU-notifier invalidation inside the kernel driver: measured 30 - 35 ms per 64 MB vector, four times the copy itself.  With the
 * staging buffers the driver never learns about the pages; the host memcpy overlaps the DMA of the previous piece. */
#define STAGE_BYTES ((size_t)16 << 20)
static struct { void *buf[2], *ev[2]; int busy[2]; } stage;
/* the staging buffers are one set per process: a thread of the program that faults on a vector (lis_pages.c) copies through them while the
 * thread that drives the library may be uploading another vector -- one copy at a time */
#include <pthread.h>
#include <unistd.h>
static pthread_mutex_t stage_lock = PTHREAD_MUTEX_INITIALIZER;

static LIS_INT stage_ready(void)
{
	if (stage.buf[0]) return LIS_SUCCESS;
	for (int k = 0; k < 2; k++) {
		HIPCHK(liship_malloc_host(&stage.buf[k], STAGE_BYTES));
		HIPCHK(liship_event_create(&stage.ev[k]));
	}
	return LIS_SUCCESS;
}

/* memcpy by the host threads (one thread moves ~10 GB/s, PCIe 5 takes 50) -- except inside the page-fault handler, which runs on whatever thread of the
 * program touched the page: no OpenMP region is opened from a signal handler (the faulting thread may sit inside a region of its own) */
static __thread int in_fault_copy;
static void copy_threads(void *dst, const void *src, size_t bytes)
{
	const size_t piece = (size_t)1 << 20;
	const long long pieces = (long long)((bytes + piece - 1) / piece);
	if (pieces < 4 || in_fault_copy) { memcpy(dst, src, bytes); return; }
	const int T = lisi_host_threads();
#pragma omp parallel for num_threads(T) schedule(static)
	for (long long i = 0; i < pieces; i++) {
		const size_t off = (size_t)i * piece, len = bytes - off < piece ? bytes - off : piece;
		memcpy((char *)dst + off, (const char *)src + off, len);
	}
}

static LIS_INT staged_h2d_locked(void *dst, const void *src, size_t bytes)
{
	LISCHK(stage_ready());
	int k = 0;
	for (size_t off = 0; off < bytes; off += STAGE_BYTES, k ^= 1) {
		const size_t len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
		if (stage.busy[k]) { HIPCHK(liship_event_synchronize(stage.ev[k])); stage.busy[k] = 0; }
		copy_threads(stage.buf[k], (const char *)src + off, len);
		HIPCHK(liship_memcpy_h2d((char *)dst + off, stage.buf[k], len, lisg.stream));
		HIPCHK(liship_event_record(stage.ev[k], lisg.stream));
		stage.busy[k] = 1;
	}
	return LIS_SUCCESS;
}
static LIS_INT staged_h2d(void *dst, const void *src, size_t bytes)
{
	pthread_mutex_lock(&stage_lock);
	const LIS_INT err = staged_h2d_locked(dst, src, bytes);
	pthread_mutex_unlock(&stage_lock);
	return err;
}

static LIS_INT staged_d2h_locked(void *dst, const void *src, size_t bytes)
{
	LISCHK(stage_ready());
	for (int k = 0; k < 2; k++) if (stage.busy[k]) { HIPCHK(liship_event_synchronize(stage.ev[k])); stage.busy[k] = 0; }
	size_t issued = 0, landed = 0;
	int ki = 0, kl = 0;
	while (landed < bytes) {
		while (issued < bytes && issued - landed < 2 * STAGE_BYTES) {          /* keep both buffers in flight */
			const size_t len = bytes - issued < STAGE_BYTES ? bytes - issued : STAGE_BYTES;
			HIPCHK(liship_memcpy_d2h(stage.buf[ki], (const char *)src + issued, len, lisg.stream));
			HIPCHK(liship_event_record(stage.ev[ki], lisg.stream));
			issued += len; ki ^= 1;
		}
		const size_t len = bytes - landed < STAGE_BYTES ? bytes - landed : STAGE_BYTES;
		HIPCHK(liship_event_synchronize(stage.ev[kl]));
		copy_threads((char *)dst + landed, stage.buf[kl], len);
		landed += len; kl ^= 1;
	}
	return LIS_SUCCESS;
}
/* HBM -> host memory the runtime must not learn about (page-protected arrays, through their alias mapping: lis_pages.c) */
LIS_INT lisd_staged_d2h(void *dst, const void *src, size_t bytes)
{
	pthread_mutex_lock(&stage_lock);
	const LIS_INT err = staged_d2h_locked(dst, src, bytes);
	pthread_mutex_unlock(&stage_lock);
	return err;
}

/* The same copy from INSIDE the page-fault handler (lis_pages.c on_fault), i.e. on a thread of the program, possibly while the thread that drives the library is
 * enqueueing work: a stream, two pinned buffers and two events of its own (made when the handler is installed, never inside it), so that nothing is enqueued on
 * the library's stream from a foreign thread -- a stream under hipGraph capture (LIS_AMD_GRAPHS=1) would have its capture invalidated -- and the driving
 * thread's staging buffers are never waited for.  What the data depends on has been enqueued on lisg.stream by the time its pages lost their access: the copy
 * waits for an event recorded there... by THIS thread, which is legal on a capturing stream only outside the capture, so a fault during a capture waits for
 * the capture to end first (a batch of launches: microseconds).  Plain memcpy, no OpenMP. */
static struct { void *stream, *buf[2], *ev[2], *order; int ready; } fstage;
static volatile int capture_active;
static pthread_t capture_thread;                                    /* the thread that began the capture in progress */
static pthread_mutex_t fault_lock = PTHREAD_MUTEX_INITIALIZER;      /* one fault-time copy at a time; a capture begins only between two of them */
void lisd_capture_mark(int on)
{
	if (on) { pthread_mutex_lock(&fault_lock); capture_thread = pthread_self(); capture_active = 1; pthread_mutex_unlock(&fault_lock); }
	else capture_active = 0;
}
LIS_INT lisd_fault_stage_prepare(void)
{
	if (fstage.ready || !lisg.device_ready) return LIS_SUCCESS;
	HIPCHK(liship_stream_create(&fstage.stream));
	for (int k = 0; k < 2; k++) {
		HIPCHK(liship_malloc_host(&fstage.buf[k], STAGE_BYTES));
		HIPCHK(liship_event_create(&fstage.ev[k]));
	}
	HIPCHK(liship_event_create(&fstage.order));
	fstage.ready = 1;
	return LIS_SUCCESS;
}
LIS_INT lisd_staged_d2h_fault(void *dst, const void *src, size_t bytes)
{
	if (!fstage.ready) {               /* (the handler was installed before the device existed: the shared path -- with plain memcpy: no OpenMP region from inside a signal handler) */
		in_fault_copy = 1;
		const LIS_INT e = lisd_staged_d2h(dst, src, bytes);
		in_fault_copy = 0;
		return e;
	}
	LIS_INT err = LIS_SUCCESS;
	pthread_mutex_lock(&fault_lock);
	/* a capture in progress is waited out -- a batch of launches, microseconds.  Two ways this could never end, both loud instead of a silent spin:
	 * the capturing thread ITSELF faulted (nobody is left to end the capture), or the capture was abandoned on an error path without lisd_capture_mark(0) */
	if (capture_active && pthread_equal(capture_thread, pthread_self())) {
		fprintf(stderr, "liblis_amd: the thread that is capturing a hipGraph touched a page-protected array whose data lives in HBM: the copy cannot be ordered inside the capture -- aborting\n");
		abort();
	}
	for (int waited_us = 0; capture_active; waited_us += 50) {
		if (waited_us > 10 * 1000 * 1000) {
			fprintf(stderr, "liblis_amd: a hipGraph capture has not ended for 10 s while a page fault waits for it: taking the capture for abandoned\n");
			capture_active = 0;
			break;
		}
		usleep(50);
	}
	int rc = liship_event_record(fstage.order, lisg.stream);
	if (!rc) rc = liship_stream_wait_event(fstage.stream, fstage.order);
	in_fault_copy = 1;
	size_t issued = 0, landed = 0;
	int ki = 0, kl = 0;
	while (!rc && landed < bytes) {
		while (!rc && issued < bytes && issued - landed < 2 * STAGE_BYTES) {
			const size_t len = bytes - issued < STAGE_BYTES ? bytes - issued : STAGE_BYTES;
			rc = liship_memcpy_d2h(fstage.buf[ki], (const char *)src + issued, len, fstage.stream);
			if (!rc) rc = liship_event_record(fstage.ev[ki], fstage.stream);
			issued += len; ki ^= 1;
		}
		if (rc) break;
		const size_t len = bytes - landed < STAGE_BYTES ? bytes - landed : STAGE_BYTES;
		rc = liship_event_synchronize(fstage.ev[kl]);
		if (!rc) memcpy((char *)dst + landed, fstage.buf[kl], len);
		landed += len; kl ^= 1;
	}
	in_fault_copy = 0;
	if (rc) { (void)liship_stream_synchronize(fstage.stream); err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	pthread_mutex_unlock(&fault_lock);
	return err;
}

/* host -> HBM when the host side is the truth.  COHERENT with page protection (the default, lis_pages.c): the host array was written
 * since the last upload exactly when its pages are read + write (dev_valid == 0); eager COHERENT: always (nothing tells) */
LIS_INT lisd_vec_in(LIS_VECTOR v, double **out)
{
	lisd_vec *d = VDEV(v);
	LISCHK(lisd_vec_reserve(v, vec_len(v)));
	const int lazy = lisp_lazy() && d->region;
	const int host_is_truth = (lisg.residency == LIS_AMD_COHERENT && !lazy) ? (d->host_valid || !d->dev_valid) : !d->dev_valid;
	if (host_is_truth && v->value) {
		size_t len = d->hlen < d->cap ? d->hlen : d->cap;
		if (lazy) LISCHK(staged_h2d(d->d, v->value, len * sizeof(double)));          /* (the array has been read when this returns) */
		else HIPCHK(liship_memcpy_h2d(d->d, v->value, len * sizeof(double), lisg.stream));
		d->dev_valid = 1;
		if (lisg.residency == LIS_AMD_COHERENT) d->host_valid = 1;
		if (lazy) {
			lisp_protect(v, LISP_RO);        /* both sides agree: a host write from here on faults and marks the HBM copy stale */
			/* the protection could not be had (mprotect out of VMAs, no handler): nothing would report a host write, so the HBM copy
			 * counts as stale again and the next call uploads afresh -- slower, never wrong */
			if (lisp_state(v) != LISP_RO) d->dev_valid = 0;
		}
	}
	*out = d->d;
	return LIS_SUCCESS;
}

LIS_INT lisd_vec_out(LIS_VECTOR v, double **out)
{
	LISCHK(lisd_vec_reserve(v, vec_len(v)));
	*out = VDEV(v)->d;
	return LIS_SUCCESS;
}

LIS_INT lisd_vec_done(LIS_VECTOR v)
{
	lisd_vec *d = VDEV(v);
	d->dev_valid = 1;
	d->host_valid = 0;
	if (lisg.residency == LIS_AMD_COHERENT) {
		if (lisp_lazy() && d->region) { lisp_protect(v, LISP_NONE); if (lisp_state(v) == LISP_NONE) return LIS_SUCCESS; }   /* the first host access brings it home */
		return lisd_vec_to_host(v);
	}
	return LIS_SUCCESS;
}

LIS_INT lisd_vec_to_host(LIS_VECTOR v)
{
	lisd_vec *d = VDEV(v);
	if (lisp_lazy() && d->region) return lisp_vec_home(v);      /* through the alias mapping: the program's pages open when the data is there */
	if (d->host_valid || !d->dev_valid || !d->d || !v->value) {
		d->host_valid = 1;
		lisp_protect(v, LISP_RW);                 /* (pages of a vector protected before the mode was switched to eager) */
		return LIS_SUCCESS;
	}
	size_t len = d->hlen < d->cap ? d->hlen : d->cap;
	lisp_protect(v, LISP_RW);
	HIPCHK(liship_memcpy_d2h(v->value, d->d, len * sizeof(double), lisg.stream));
	HIPCHK(liship_stream_synchronize(lisg.stream));
	d->host_valid = 1;
	return LIS_SUCCESS;
}

/* the library itself is about to write value[] on the host: current data first when `keep`, pages writable, HBM copy stale */
LIS_INT lisd_vec_host_write(LIS_VECTOR v, int keep)
{
	lisd_vec *d = VDEV(v);
	if (keep && !d->host_valid) LISCHK(lisd_vec_to_host(v));
	lisp_protect(v, LISP_RW);
	d->host_valid = 1;
	d->dev_valid = 0;
	return LIS_SUCCESS;
}

void lisd_vec_free(LIS_VECTOR v)
{
	lisd_vec *d = VDEV(v);
	if (d->d) { (void)liship_free(d->d); d->d = NULL; d->cap = 0; }
	d->dev_valid = 0;
}

LIS_INT lis_amd_vector_sync_host(LIS_VECTOR v) { return lisd_vec_to_host(v); }
LIS_INT lis_amd_vector_host_modified(LIS_VECTOR v) { lisp_protect(v, LISP_RW); VDEV(v)->host_valid = 1; VDEV(v)->dev_valid = 0; return LIS_SUCCESS; }
LIS_INT lis_amd_vector_device_modified(LIS_VECTOR v) { VDEV(v)->host_valid = 0; VDEV(v)->dev_valid = 1; if (lisp_lazy()) lisp_protect(v, LISP_NONE); return LIS_SUCCESS; }
LIS_INT lis_amd_vector_device_ptr(LIS_VECTOR v, LIS_SCALAR **dptr) { return lisd_vec_in(v, dptr); }

/* ------------------------------------------------------------------ matrices */
static LIS_INT up_i(int **dst, const int *src, size_t count)
{
	void *p = NULL;
	HIPCHK(lisd_malloc(&p, (count + 4) * sizeof(int)));          /* +4: 16 B slack for vector loads */
	if (count) HIPCHK(liship_memcpy_h2d(p, src, count * sizeof(int), lisg.stream));
	*dst = (int *)p;
	return LIS_SUCCESS;
}
static LIS_INT up_d(double **dst, const double *src, size_t count)
{
	void *p = NULL;
	HIPCHK(lisd_malloc(&p, (count + 2) * sizeof(double)));
	if (count) HIPCHK(liship_memcpy_h2d(p, src, count * sizeof(double), lisg.stream));
	*dst = (double *)p;
	return LIS_SUCCESS;
}

/* XCD strips of the native ELL / DIA kernels (liship_spmv_formats_set_plane): the plane of the grid is the largest offset most rows reach -- for DIA read off its
 * offsets (the largest positive one: a diagonal serves every row it fits), for ELL found by two passes over index[] in HBM.  Used only where x does not fit the
 * 256 MB Infinity Cache (measured: 512^3 ELL 0.671 -> 0.733 of the roofline with the counters' traffic at 1.01 x the algorithmic bytes instead of 1.26 x, DIA 0.687 ->
 * 0.709; at 256^3, where x + y sit in that cache, the strips COST ELL 7 %: profiles/r06_formats_512.txt). */
static void fmt_find_plane(lisd_mat *d, const int *host_dia_offsets)
{
	d->xs_rows = 0;
	if (d->type == LIS_MATRIX_DIA && host_dia_offsets) {
		int best = 0;
		for (int k = 0; k < d->nnd; k++) if (host_dia_offsets[k] > best && host_dia_offsets[k] < d->n) best = host_dia_offsets[k];
		d->xs_rows = best;
	} else if (d->type == LIS_MATRIX_ELL && d->index) {
		int plane = 0;
		if (liship_ell_scan_band(d->n, d->maxnzr, d->index, &plane, lisg.stream) == 0) d->xs_rows = plane;
	}
}
static void fmt_strips(const lisd_mat *d)        /* in front of every whole-matrix launch of a native ELL / DIA kernel */
{
	(void)liship_spmv_formats_set_plane((d->xs_rows > 0 && (size_t)d->n * sizeof(double) > ((size_t)256 << 20)) ? d->xs_rows : 0);
}

/* the permutation the last reordered plan found (liship_csr_plan_reorder), tried first by the next plan of the same size: a program that edits A->value between solves
 * rebuilds the HBM copy and its plan each time, and the walk (1.4 s on the Queen-class matrix) is most of that.  One entry; a hint is only ever a hint. */
static struct { int *perm; int n; long long nnz; } renum_cache;
static void renum_cache_drop(void) { free(renum_cache.perm); renum_cache.perm = NULL; renum_cache.n = 0; renum_cache.nnz = 0; }
/* rows of the local matrix that reference no ghost column, as one maximal run [b,e) */
/* the row split of a CSR-ordered HBM matrix and, where its columns allow it, the one-byte column codes
 * (liship.h "index coding"; LIS_AMD_NO_INDEX_CODES=1 keeps the 4 B indices for A/B measurements) */
static LIS_INT csr_plan_impl(liship_csr_plan_t *plan, int n, int ncols, const int *dptr, const int *dindex, const double *dvalue, int reorder);
LIS_INT lisd_csr_plan(liship_csr_plan_t *plan, int n, const int *dptr, const int *dindex, const double *dvalue) { return csr_plan_impl(plan, n, 0, dptr, dindex, dvalue, 1); }
/* ... of a rank's local rows: columns [n, ncols) are its ghost columns (the renumbered form keeps them apart: liship_csr_plan_set_ghost_columns) */
LIS_INT lisd_csr_plan_cols(liship_csr_plan_t *plan, int n, int ncols, const int *dptr, const int *dindex, const double *dvalue) { return csr_plan_impl(plan, n, ncols, dptr, dindex, dvalue, 1); }
/* ... of a matrix no solve iterates on (a transposed copy, a scaled copy, the halves of a split JAD matrix): no renumbered form (products would not use it) */
LIS_INT lisd_csr_plan_plain(liship_csr_plan_t *plan, int n, const int *dptr, const int *dindex, const double *dvalue) { return csr_plan_impl(plan, n, 0, dptr, dindex, dvalue, 0); }
/* the renumbered form of a plan (liship_csr_plan_reorder: never an error when the matrix does not qualify; out of memory leaves the plan as it was) */
static LIS_INT plan_try_reorder(liship_csr_plan_t plan, int n, const int *dptr, const int *dindex, const double *dvalue)
{
	long long pnnz = 0;
	(void)liship_csr_plan_info(plan, NULL, &pnnz, NULL);
	const int *hint = (renum_cache.perm && renum_cache.n == n && renum_cache.nnz == pnnz) ? renum_cache.perm : NULL;      /* the last walk, when the sizes match (a matrix whose values were edited; any other matrix drops it for a walk of its own) */
	int rc = liship_csr_plan_reorder_with(plan, dptr, dindex, dvalue, 0, hint, lisg.stream);
	if (rc && rc != 2) HIPCHK(rc);
	if (!rc && liship_csr_plan_reordered(plan) > 0 && !hint) {
		int *keep = (int *)malloc(sizeof(int) * (size_t)n);
		if (keep && liship_csr_plan_reorder_permutation(plan, keep) == 0) {
			free(renum_cache.perm);
			renum_cache.perm = keep; renum_cache.n = n; renum_cache.nnz = pnnz;
		} else free(keep);
	}
	return LIS_SUCCESS;
}
/* LAZY renumbering (round 6): called by lis_solve before it looks for a renumbered form.  A CSR copy on one rank whose plan has served lisg.reorder_after products
 * in the caller's numbering gets the attempt once; what the attempt costs (the numbering found on the device -- kernels/csr_order.hpp --, P A P^T and its plan built in HBM: +0.17 s and +3.5 GB on the
 * Queen-class matrix; rounds 4-5 walked the graph on the host: 1.6 s) is paid by a program that has shown it iterates long enough to earn it back (0.06-0.1 ms per
 * iteration there: ~3000 iterations), never by the first solves. */
LIS_INT lisd_mat_lazy_reorder(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	if (lisg.no_reorder || lisg.reorder_after <= 0 || !d->ready || d->reorder_tried) return LIS_SUCCESS;
	{	/* when: the ski-rental point -- build once the products served have cost about what the form costs.  Lists that exist but are long (the Queen class: the form
		 * saves 5-15 % of an iteration) wait for reorder_after products; a plan whose lists FAILED (no locality at all: 30-40 % of the roofline, the form doubles the
		 * rate; building it costs ~200 of those products whatever the size) waits for a sixteenth of that (256 by default) */
		long long wait = lisg.reorder_after;
		if (d->type == LIS_MATRIX_CSR && d->plan && liship_csr_plan_lists_failed(d->plan)) wait = wait / 16 > 0 ? wait / 16 : 1;
		if (d->served < wait) return LIS_SUCCESS;
	}
	/* (several ranks: each renumbers its own rows and owned columns -- the plan knows its ghost columns --; a matrix served as CSR from another layout with ghost columns stays as it is) */
	if (d->type != LIS_MATRIX_CSR || !d->plan || !d->value || d->split_jad || d->solve_holds || A->is_scaled || A->is_splited || d->n != A->n ||
	    (A->np != A->n && A->matrix_type != LIS_MATRIX_CSR)) return LIS_SUCCESS;
	d->reorder_tried = 1;
	return plan_try_reorder(d->plan, d->n, d->ptr, d->index, d->value);
}
LIS_INT lis_amd_set_reorder_after(long long products) { lisg.reorder_after = products < 0 ? 0 : products; return LIS_SUCCESS; }
long long lis_amd_matrix_products_served(LIS_MATRIX A) { return MDEV(A)->served; }

static LIS_INT csr_plan_impl(liship_csr_plan_t *plan, int n, int ncols, const int *dptr, const int *dindex, const double *dvalue, int reorder)
{
	int rc = liship_csr_plan_create(plan, n, dptr, lisg.stream);
	if (rc && lis_amd_trim_count() > 0) rc = liship_csr_plan_create(plan, n, dptr, lisg.stream);   /* the plan allocates in the kernel layer */
	HIPCHK(rc);
	if (ncols > n) HIPCHK(liship_csr_plan_set_ghost_columns(*plan, ncols));
	if (!lisg.no_index_codes) {
		/* the codes are an optimisation: a matrix that cannot have them (out of memory included) keeps its 4 B indices */
		rc = liship_csr_plan_encode_indices(*plan, dptr, dindex, lisg.stream);
		if (rc && lis_amd_trim_count() > 0) rc = liship_csr_plan_encode_indices(*plan, dptr, dindex, lisg.stream);
		if (rc && rc != 2 /* hipErrorOutOfMemory */) HIPCHK(rc);
		if (!rc && !lisg.no_row_patterns && liship_csr_plan_coded(*plan)) {       /* whole rows that repeat: one byte per row */
			rc = liship_csr_plan_encode_row_patterns(*plan, dptr, lisg.stream);
			if (rc && rc != 2) HIPCHK(rc);
			if (!rc && !lisg.no_value_records && dvalue) {        /* ... and carry the same values: nothing left to stream */
				rc = liship_csr_plan_encode_row_values(*plan, dptr, dvalue, lisg.stream);
				if (rc && rc != 2) HIPCHK(rc);
			}
		}
	}
	if (!lisg.no_local_columns && !liship_csr_plan_coded(*plan)) {      /* block-local columns where they pay: long rows, and (round 6) short rows whose row blocks share their columns */
		/* LIS_AMD_NO_INDEX_CODES=1 asks for the reference's own arrays in the product of a short-row matrix (the contract form): no lists for short rows then either */
		HIPCHK(liship_spmv_csr_set_local_short_rows((lisg.no_index_codes || lisg.no_local_short_rows) ? 0 : 1));
		rc = liship_csr_plan_localize_columns(*plan, dptr, dindex, lisg.stream);
		if (rc && lis_amd_trim_count() > 0) rc = liship_csr_plan_localize_columns(*plan, dptr, dindex, lisg.stream);
		if (rc && rc != 2) HIPCHK(rc);
		/* lists that stay long say the numbering has no locality: rows and columns renumbered inside the plan (one rank: its row ranges follow the original order).
		 * At plan time only when asked (LIS_AMD_REORDER_AFTER=0); by default the plan first serves lisg.reorder_after products in the caller's numbering
		 * (lisd_mat_lazy_reorder): building the form costs ~3000 iterations of what it saves per iteration on the Queen-class matrix, and the solves
		 * people time first take 40-50 */
		if (!rc && reorder && !lisg.no_reorder && (lisg.nprocs == 1 || ncols >= n) && dvalue && lisg.reorder_after == 0) LISCHK(plan_try_reorder(*plan, n, dptr, dindex, dvalue));
	}
	/* a plan that streams index[] / codes (no row patterns): the plane of a structured grid from the band of the matrix, for the XCD strips */
	rc = liship_csr_plan_scan_band(*plan, dptr, dindex, lisg.stream);
	if (rc && rc != 2) HIPCHK(rc);
	return LIS_SUCCESS;
}

/* the longest run of rows that reference no ghost column (columns >= n): those rows run while the halo is in flight, the boundary
 * rows after it.  Read from the host layout of whatever format A has (CSR / CSC / ELL / DIA / JAD; BSR and split matrices: none). */
static void find_inner_rows(LIS_MATRIX A, int *b, int *e)
{
	const int n = A->n;
	*b = 0; *e = 0;
	if (A->np == n) { *e = A->matrix_type == LIS_MATRIX_BSR ? A->nr : n; return; }       /* no ghost columns at all (BSR counts block rows) */
	if (A->is_splited || n <= 0) return;
	unsigned char *ghost = (unsigned char *)calloc((size_t)n + 1, 1);
	if (!ghost) return;                                               /* (no overlap then: exchange first) */
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		if (!A->ptr) { free(ghost); return; }
		for (int r = 0; r < n; r++)
			for (int k = A->ptr[r]; k < A->ptr[r + 1]; k++) if (A->index[k] >= n) { ghost[r] = 1; break; }
		break;
	case LIS_MATRIX_CSC:
		for (int c = n; c < A->np; c++)
			for (int k = A->ptr[c]; k < A->ptr[c + 1]; k++) ghost[A->index[k]] = 1;
		break;
	case LIS_MATRIX_ELL:
		for (int j = 0; j < A->maxnzr; j++)
			for (int r = 0; r < n; r++) if (A->index[(size_t)j * n + r] >= n) ghost[r] = 1;
		break;
	case LIS_MATRIX_DIA:                                              /* a diagonal reaches the ghosts in the rows where n <= r + offset < np (explicit zeros are read too) */
		for (int dgl = 0; dgl < A->nnd; dgl++) {
			const long long o = A->index[dgl];
			long long lo = (long long)n - o, hi = (long long)A->np - o;
			if (lo < 0) lo = 0;
			if (hi > n) hi = n;
			for (long long r = lo; r < hi; r++) ghost[r] = 1;
		}
		break;
	case LIS_MATRIX_JAD:
		for (int j = 0; j < A->maxnzr; j++)
			for (int sl = 0; sl < A->ptr[j + 1] - A->ptr[j]; sl++) if (A->index[A->ptr[j] + sl] >= n) ghost[A->row[sl]] = 1;
		break;
	case LIS_MATRIX_BSR: {                                            /* in BLOCK rows: ghost columns start on a fresh block column (lis_matrix_bsr.c:425-428) */
		if (!A->bptr) { free(ghost); return; }
		const int first_ghost = (n + A->bnc - 1) / A->bnc;
		for (int br = 0; br < A->nr; br++)
			for (int k = A->bptr[br]; k < A->bptr[br + 1]; k++) if (A->bindex[k] >= first_ghost) { ghost[br] = 1; break; }
		ghost[A->nr] = 1;
		int bb = 0, be = 0, rb = 0;
		for (int r = 0; r <= A->nr; r++)
			if (ghost[r]) { if (r - rb > be - bb) { bb = rb; be = r; } rb = r + 1; }
		free(ghost);
		*b = bb; *e = be;
		return;
	}
	default:
		free(ghost);
		return;
	}
	ghost[n] = 1;
	int best_b = 0, best_e = 0, run_b = 0;
	for (int r = 0; r <= n; r++)
		if (ghost[r]) {
			if (r - run_b > best_e - best_b) { best_b = run_b; best_e = r; }
			run_b = r + 1;
		}
	free(ghost);
	*b = best_b; *e = best_e;
}

static LIS_INT upload_csc_as_csr(LIS_MATRIX A, lisd_mat *d)
{
	const int n = A->n, np = A->np, nnz = A->nnz;
	int *tptr = (int *)calloc((size_t)n + 2, sizeof(int));
	int *tidx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
	double *tval = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
	if (!tptr || !tidx || !tval) { free(tptr); free(tidx); free(tval); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "csc transpose\n"); }
	for (int k = 0; k < nnz; k++) tptr[A->index[k] + 1]++;
	for (int r = 0; r < n; r++) tptr[r + 1] += tptr[r];
	int *fill = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
	memcpy(fill, tptr, sizeof(int) * (size_t)n);
	for (int c = 0; c < np; c++)                    /* columns ascending: the reference's summation order */
		for (int k = A->ptr[c]; k < A->ptr[c + 1]; k++) {
			const int dst = fill[A->index[k]]++;
			tidx[dst] = c; tval[dst] = A->value[k];
		}
	free(fill);
	LIS_INT err = up_i(&d->ptr, tptr, (size_t)n + 1);
	if (!err) err = up_i(&d->index, tidx, (size_t)nnz);
	if (!err) err = up_d(&d->value, tval, (size_t)nnz);
	if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	free(tptr); free(tidx); free(tval);
	return err;
}

/* A JAD matrix is laid out in HBM row by row, in the ORIGINAL row order: row perm[s] gets the s-th entry of every
 * jagged diagonal that is long enough, diagonal by diagonal -- the order in which lis_matvec_jad adds them to
 * y[perm[s]] starting from 0 (lis_matvec_jad.c:57-75), so the CSR kernel forms the same sums bit for bit.  The
 * jagged layout is what a vector CPU wants; on MI355X it costs a permuted y and maxnzr separate streams per lane
 * (68 % of the roofline, spmv_jad_kernel, kept for the kernel-level API), the row layout runs at the CSR rate. */
static LIS_INT upload_jad_as_csr(LIS_MATRIX A, lisd_mat *d)
{
	const int n = A->n, nnz = A->nnz, maxnzr = A->maxnzr;
	int *cptr = (int *)calloc((size_t)n + 2, sizeof(int));
	int *cidx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
	double *cval = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
	int *fill = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
	if (!cptr || !cidx || !cval || !fill) { free(cptr); free(cidx); free(cval); free(fill); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "jad re-layout\n"); }
	for (int j = 0; j < maxnzr; j++) {
		const int len = A->ptr[j + 1] - A->ptr[j];
		for (int s = 0; s < len; s++) cptr[A->row[s] + 1]++;
	}
	for (int r = 0; r < n; r++) cptr[r + 1] += cptr[r];
	memcpy(fill, cptr, sizeof(int) * (size_t)n);
	for (int j = 0; j < maxnzr; j++) {                /* diagonals ascending: each row receives its entries in summation order */
		const int b = A->ptr[j], len = A->ptr[j + 1] - b;
		for (int s = 0; s < len; s++) {
			const int dst = fill[A->row[s]]++;
			cidx[dst] = A->index[b + s]; cval[dst] = A->value[b + s];
		}
	}
	free(fill);
	LIS_INT err = up_i(&d->ptr, cptr, (size_t)n + 1);
	if (!err) err = up_i(&d->index, cidx, (size_t)nnz);
	if (!err) err = up_d(&d->value, cval, (size_t)nnz);
	if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	free(cptr); free(cidx); free(cval);
	return err;
}

/* ELL and DIA matrices with constant coefficients.  Both formats add the terms of a row in a fixed order from 0 -- ELL its maxnzr
 * slots, padding included (value 0, index i: lis_matvec_ell.c:113-128), DIA its diagonals in ascending order over the rows where
 * i + offset stays inside the matrix, explicit zeros included (lis_matvec_dia.c:148-172) -- so a CSR layout that lists exactly
 * those terms, zeros and all, in that order, gives the CSR kernel the same sums bit for bit (0 * x[i] is kept: it is NaN when x[i]
 * is not finite, as in the reference).  The layout pays when the plan then finds value records (liship.h): a constant-coefficient
 * stencil in ELL or DIA streams 100 or 72 B per row, its row form one byte per row.  A cheap screen (few distinct values among the
 * first entries) keeps every other matrix away from the attempt; when the plan finds no value records the row form is dropped and
 * the native arrays are uploaded as before.  *taken says which. */
static int few_distinct_values(const double *v, size_t count)
{
	unsigned long long seen[8];
	int ns = 0;
	const size_t lim = count < 65536 ? count : 65536;
	for (size_t k = 0; k < lim; k++) {
		unsigned long long b;
		memcpy(&b, v + k, 8);
		int j = 0;
		while (j < ns && seen[j] != b) j++;
		if (j == ns) { if (ns == 8) return 0; seen[ns++] = b; }
	}
	return 1;
}

static LIS_INT try_row_form(LIS_MATRIX A, lisd_mat *d, int *taken)
{
	*taken = 0;
	const int n = A->n;
	if (lisg.no_row_form || lisg.no_value_records || lisg.no_row_patterns || lisg.no_index_codes || n <= 0) return LIS_SUCCESS;
	const int width = A->matrix_type == LIS_MATRIX_ELL ? A->maxnzr : A->nnd;
	if (width < 1 || width > 32 || (long long)n * width >= 0x7fffffffLL) return LIS_SUCCESS;    /* value records hold up to 32 entries per row */
	if (!few_distinct_values(A->value, (size_t)n * (size_t)width)) return LIS_SUCCESS;
	int *cptr = (int *)malloc(sizeof(int) * ((size_t)n + 1));
	if (!cptr) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "row form\n");
	const int T = lisi_host_threads();
	const long long ncols = A->np;
	if (A->matrix_type == LIS_MATRIX_ELL) {
		for (int i = 0; i <= n; i++) cptr[i] = i * width;
	} else {
		cptr[0] = 0;
		for (int i = 0; i < n; i++) {                 /* the diagonals that reach row i: 0 <= i + offset < np (lis_matvec_dia.c:154-160) */
			int c = 0;
			for (int k = 0; k < width; k++) { const long long j = (long long)i + A->index[k]; c += (j >= 0 && j < ncols); }
			cptr[i + 1] = cptr[i] + c;
		}
	}
	const size_t nnz = (size_t)cptr[n];
	int *cidx = (int *)malloc(sizeof(int) * (nnz ? nnz : 1));
	double *cval = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
	if (!cidx || !cval) { free(cptr); free(cidx); free(cval); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "row form\n"); }
	if (A->matrix_type == LIS_MATRIX_ELL) {
#pragma omp parallel for num_threads(T) schedule(static)
		for (int i = 0; i < n; i++)
			for (int j = 0; j < width; j++) { cidx[(size_t)i * width + j] = A->index[(size_t)j * n + i]; cval[(size_t)i * width + j] = A->value[(size_t)j * n + i]; }
	} else {
#pragma omp parallel for num_threads(T) schedule(static)
		for (int i = 0; i < n; i++) {
			int at = cptr[i];
			for (int k = 0; k < width; k++) {
				const long long j = (long long)i + A->index[k];
				if (j >= 0 && j < ncols) { cidx[at] = (int)j; cval[at] = A->value[(size_t)k * n + i]; at++; }
			}
		}
	}
	LIS_INT err = up_i(&d->ptr, cptr, (size_t)n + 1);
	if (!err) err = up_i(&d->index, cidx, nnz);
	if (!err) err = up_d(&d->value, cval, nnz);
	if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	free(cptr); free(cidx); free(cval);
	if (!err) err = lisd_csr_plan(&d->plan, n, d->ptr, d->index, d->value);
	if (!err && liship_csr_plan_value_records(d->plan)) { *taken = 1; d->type = LIS_MATRIX_CSR; d->nnz = (LIS_INT)nnz; return LIS_SUCCESS; }
	if (d->plan) { (void)liship_csr_plan_destroy(d->plan); d->plan = NULL; }          /* not this matrix: the native layout */
	(void)liship_free(d->ptr); (void)liship_free(d->index); (void)liship_free(d->value);
	d->ptr = NULL; d->index = NULL; d->value = NULL;
	if (err == LIS_ERR_OUT_OF_MEMORY) err = LIS_SUCCESS;   /* an optimisation: out of memory on the way is not an error */
	return err;
}

/* A split matrix (lis_split.c) lives in HBM as CSR rows that list the terms of a row in the order the reference's is_splited
 * branch adds them -- D x first -- and the kernels start the sum at -0.0, which makes the first product the initial value
 * (t0 = D[i]*x[i]; t0 += ...: lis_matvec_csr.c:70-87) bit for bit, signed zeros included.  JAD is not one chain:
 * (D x + sum over L) + sum over U with both partial sums started at 0 (lis_matvec_jad.c:60-140) -- two products, then two
 * element-wise passes (lisd_spmv). */
static LIS_INT upload_rows(lisd_mat *d, LIS_INT rows, LIS_INT *ptr, LIS_INT *idx, LIS_SCALAR *val, int **dptr, int **didx, double **dval, liship_csr_plan_t *plan, int from_zero, int half)
{
	const LIS_INT nnz = ptr[rows];
	LIS_INT err = up_i(dptr, ptr, (size_t)rows + 1);
	if (!err) err = up_i(didx, idx, (size_t)nnz);
	if (!err) err = up_d(dval, val, (size_t)nnz);
	if (!err) { int rc = liship_stream_synchronize(lisg.stream); if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	free(ptr); free(idx); free(val);
	if (err) return err;
	if (half) LISCHK(lisd_csr_plan_plain(plan, rows, *dptr, *didx, *dval));
	else LISCHK(lisd_csr_plan(plan, rows, *dptr, *didx, *dval));
	if (!from_zero) HIPCHK(liship_csr_plan_set_first_term_initialises(*plan, 1));
	(void)d;
	return LIS_SUCCESS;
}

static LIS_INT upload_split(LIS_MATRIX A, lisd_mat *d)
{
	LIS_INT *ptr, *idx; LIS_SCALAR *val;
	if (A->matrix_type == LIS_MATRIX_JAD) {
		LISCHK(lisi_split_jad_part(A, 0, &ptr, &idx, &val));
		LISCHK(upload_rows(d, A->n, ptr, idx, val, &d->ptr, &d->index, &d->value, &d->plan, 1, 1));
		LISCHK(lisi_split_jad_part(A, 1, &ptr, &idx, &val));
		LISCHK(upload_rows(d, A->n, ptr, idx, val, &d->u_ptr, &d->u_index, &d->u_value, &d->u_plan, 1, 1));
		LISCHK(up_d(&d->dsplit, A->D->value, (size_t)A->n));
		HIPCHK(lisd_malloc((void **)&d->jw, ((size_t)A->n + 16) * sizeof(double)));
		d->type = LIS_MATRIX_CSR;
		d->split_jad = 1;
		return LIS_SUCCESS;
	}
	LIS_INT rows; int from_zero;
	LISCHK(lisi_split_rows(A, &rows, &ptr, &idx, &val, &from_zero));
	d->nnz = ptr[rows];
	LISCHK(upload_rows(d, rows, ptr, idx, val, &d->ptr, &d->index, &d->value, &d->plan, from_zero, 0));
	d->type = LIS_MATRIX_CSR;
	d->n = rows;                          /* BSR: nr*bnr rows, the padding rows included (the vectors carry the pad) */
	return LIS_SUCCESS;
}

/* BSR matrices with constant coefficients (the 2 x 2 blocking of a stencil streams its explicit zeros: 0.38 ms at 256^3 where every other format takes 0.05): as for
 * ELL / DIA above, CSR rows that list the format's terms in the format's order -- lis_matvec_bsr adds to a scalar row its blocks in order, a block's columns in order,
 * zeros included (lis_matvec_bsr.c:123-148, :293-343) -- are built IN HBM from the native arrays (liship_bsr_to_rows) and kept when the plan finds value records on them.
 * Only without padding (n a multiple of bnr, the columns a multiple of bnc: the padded x entries would otherwise be columns of the row form) and in single-rank jobs.
 * *taken = 1: d->ptr / index / value / plan hold the row form, d->type is CSR; the native arrays stay with the caller. */
/* (the facts about the matrix come as arguments: the in-HBM conversion calls this before the target's header is filled in) */
static LIS_INT try_bsr_row_form(int n, int np, int bnr, int bnc, int splited, lisd_mat *d, const int *dbptr, const int *dbindex, const double *dbvalue, LIS_INT bnnz, int values_few, int *taken)
{
	*taken = 0;
	if (lisg.no_row_form || lisg.no_value_records || lisg.no_row_patterns || lisg.no_index_codes || lisg.nprocs > 1 || n <= 0 || bnnz <= 0 || !values_few) return LIS_SUCCESS;
	if (n % bnr != 0 || np % bnc != 0 || np != n || splited) return LIS_SUCCESS;
	const long long slots = (long long)bnnz * bnr * bnc;
	if (slots >= 0x7fffffffLL || (slots / n) > 32) return LIS_SUCCESS;          /* value records hold up to 32 entries per row */
	int *rptr = NULL, *ridx = NULL; double *rval = NULL;
	if (lisd_malloc((void **)&rptr, sizeof(int) * ((size_t)n + 5)) || lisd_malloc((void **)&ridx, sizeof(int) * ((size_t)slots + 4)) ||
	    lisd_malloc((void **)&rval, sizeof(double) * ((size_t)slots + 2))) {
		(void)liship_free(rptr); (void)liship_free(ridx); (void)liship_free(rval);
		return LIS_SUCCESS;                                  /* an optimisation: out of memory on the way is not an error */
	}
	LIS_INT err = LIS_SUCCESS;
	int rc = liship_bsr_to_rows(n, bnr, bnc, dbptr, dbindex, dbvalue, rptr, ridx, rval, lisg.stream);
	if (rc) err = lisi_hip_error(__FILE__, __func__, __LINE__, rc);
	liship_csr_plan_t plan = NULL;
	if (!err) err = lisd_csr_plan(&plan, n, rptr, ridx, rval);
	if (!err && plan && liship_csr_plan_value_records(plan)) {
		if (bnr == bnc && bnr <= 4 && liship_csr_plan_value_records(plan) == 2)       /* a lane per block row (optional; the plan decides) */
			(void)liship_csr_plan_encode_block_rows(plan, bnr, rptr, lisg.stream);
		d->ptr = rptr; d->index = ridx; d->value = rval; d->plan = plan;
		d->type = LIS_MATRIX_CSR; d->nnz = (int)slots;
		*taken = 1;
		return LIS_SUCCESS;
	}
	if (plan) (void)liship_csr_plan_destroy(plan);
	(void)liship_free(rptr); (void)liship_free(ridx); (void)liship_free(rval);
	return err == LIS_ERR_OUT_OF_MEMORY ? LIS_SUCCESS : err;
}

static LIS_INT mat_upload(LIS_MATRIX A);

/* ---- host writes to adopted arrays.  The reference adopts the caller's arrays (lis_matrix_csr.c:98-103) and reads them live on every product
 * (lis_matvec_csr.c:97-109); here the product runs on an HBM copy built once.  Arrays that came from lis_matrix_malloc_<fmt> -- or that the library made itself:
 * element-wise assembly, conversions in HBM -- live on pages of the library's (lis_pages.c): under lazy coherence they are read-only while the HBM copy lives,
 * the first host write faults, opens them and sets host_written, and the next use rebuilds the copy (arrays, plan, transposed operator).  Arrays the caller
 * malloc'ed cannot be watched: lis_amd_matrix_host_modified(A) is the contract for those, and LIS_AMD_MATRIX_CHECK=1 the debugging aid -- every use of A then
 * re-hashes its host arrays and rebuilds the copy (with one line on stderr) when they changed. */
static unsigned long long hash_words(const void *p, size_t bytes)
{
	const unsigned long long *w = (const unsigned long long *)p;
	const size_t nw = bytes / 8;
	unsigned long long h = 0x9E3779B97F4A7C15ull ^ bytes;
	#pragma omp parallel for reduction(^:h) schedule(static) num_threads(lisi_host_threads())
	for (long long c = 0; c < (long long)((nw + 4095) / 4096); c++) {
		const size_t lo = (size_t)c * 4096, hi = lo + 4096 < nw ? lo + 4096 : nw;
		unsigned long long a = 0x243F6A8885A308D3ull + (unsigned long long)c, b = 0x13198A2E03707344ull;
		for (size_t i = lo; i + 1 < hi; i += 2) { a = (a ^ w[i]) * 0x9E3779B97F4A7C15ull; b = (b ^ w[i + 1]) * 0xC2B2AE3D27D4EB4Full; }
		if ((hi - lo) & 1) a = (a ^ w[hi - 1]) * 0x9E3779B97F4A7C15ull;
		h ^= (a ^ (b >> 29) ^ (a << 17)) * 0xD6E8FEB86659FD93ull;
	}
	const unsigned char *t = (const unsigned char *)p + nw * 8;
	for (size_t i = 0; i < bytes % 8; i++) h = (h ^ t[i]) * 0x100000001B3ull;
	return h;
}

static int host_arrays(LIS_MATRIX A, const void *arr[6], size_t bytes[6])
{
	const size_t n = (size_t)A->n;
	int k = 0;
#define ARR(p, b) do { if (p) { arr[k] = (p); bytes[k] = (b); k++; } } while (0)
	if (A->is_splited) return 0;                       /* (the split parts L, U, D are the library's own work arrays: not watched) */
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR: ARR(A->ptr, 4 * (n + 1)); ARR(A->index, 4 * (size_t)A->nnz); ARR(A->value, 8 * (size_t)A->nnz); break;
	case LIS_MATRIX_CSC: ARR(A->ptr, 4 * ((size_t)A->np + 1)); ARR(A->index, 4 * (size_t)A->nnz); ARR(A->value, 8 * (size_t)A->nnz); break;
	case LIS_MATRIX_ELL: ARR(A->index, 4 * n * (size_t)A->maxnzr); ARR(A->value, 8 * n * (size_t)A->maxnzr); break;
	case LIS_MATRIX_DIA: ARR(A->index, 4 * (size_t)A->nnd); ARR(A->value, 8 * n * (size_t)A->nnd); break;
	case LIS_MATRIX_JAD: ARR(A->row, 4 * n); ARR(A->ptr, 4 * ((size_t)A->maxnzr + 1)); ARR(A->index, 4 * (size_t)A->nnz); ARR(A->value, 8 * (size_t)A->nnz); break;
	case LIS_MATRIX_BSR: ARR(A->bptr, 4 * ((size_t)A->nr + 1)); ARR(A->bindex, 4 * (size_t)A->bnnz); ARR(A->value, 8 * (size_t)A->bnnz * (size_t)A->bnr * (size_t)A->bnc); break;
	default: break;
	}
#undef ARR
	return k;
}

static unsigned long long host_arrays_hash(LIS_MATRIX A)
{
	const void *arr[6]; size_t bytes[6];
	const int k = host_arrays(A, arr, bytes);
	unsigned long long h = 0;
	for (int i = 0; i < k; i++) h = (h * 0x9E3779B97F4A7C15ull) ^ hash_words(arr[i], bytes[i]);
	return h;
}

LIS_INT lisd_mat_ready(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	if (d->ready && d->solve_holds) return LIS_SUCCESS;       /* a solve in the plan's numbering has P A P^T's arrays in d->ptr / index / value (lis_solver.c): the copy stays as it is until the solve hands it back */
	if (d->ready && !d->device_only) {
		if (d->host_written) lisd_mat_free(A);             /* a host write to one of its arrays was seen (page fault): the copy is stale */
		else if (lisg.matrix_check && d->checked && lisp_lazy_arrays(A) == 0 && host_arrays_hash(A) != d->host_hash) {
			fprintf(stderr, "liblis_amd: LIS_AMD_MATRIX_CHECK: the host arrays of matrix %p changed since its HBM copy was built and lis_amd_matrix_host_modified() was "
			                "not called: rebuilding the copy\n", (void *)A);
			lisd_mat_free(A);
		}
	}
	if (d->ready) return LIS_SUCCESS;
	const LIS_INT err = mat_upload(A);
	if (err) { lisd_mat_free(A); return err; }      /* a half-made HBM copy (arrays up, plan failed ...) must not be uploaded over by the next call */
	{	/* the arrays the copy was built from: watched from here on where they live on the library's pages */
		const void *arr[6]; size_t bytes[6];
		const int k = host_arrays(A, arr, bytes);
		for (int i = 0; i < k; i++) (void)lisp_adopt(A, (void *)arr[i]);
		(void)lisp_matrix_protect(A);
		if (lisg.matrix_check) { d->host_hash = host_arrays_hash(A); d->checked = 1; }
	}
	return LIS_SUCCESS;
}

static LIS_INT mat_upload(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	LISCHK(lisd_init());
	if (A->status < LIS_MATRIX_CSR) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is not assembled\n");
	LISCHK(lisp_fill_matrix(A));
	d->host_written = 0;
	d->n = A->n; d->np = A->np; d->nnz = A->nnz;
	d->type = A->matrix_type;
	const size_t n = (size_t)A->n;
	if (A->is_splited && !(A->matrix_type == LIS_MATRIX_BSR && A->bnr != A->bnc)) {
		LISCHK(upload_split(A, d));
		HIPCHK(liship_stream_synchronize(lisg.stream));
		find_inner_rows(A, &d->inner_begin, &d->inner_end);      /* (a split matrix with ghost columns: exchange first, no overlap) */
		d->ready = 1;
		return LIS_SUCCESS;
	}
	switch (A->matrix_type) {
	case LIS_MATRIX_CSR:
		LISCHK(up_i(&d->ptr, A->ptr, n + 1));
		LISCHK(up_i(&d->index, A->index, (size_t)A->nnz));
		LISCHK(up_d(&d->value, A->value, (size_t)A->nnz));
		LISCHK(lisd_csr_plan_cols(&d->plan, A->n, A->np, d->ptr, d->index, d->value));
		break;
	case LIS_MATRIX_CSC:
		LISCHK(upload_csc_as_csr(A, d));
		d->type = LIS_MATRIX_CSR;
		LISCHK(lisd_csr_plan(&d->plan, A->n, d->ptr, d->index, d->value));
		break;
	case LIS_MATRIX_ELL:
		d->maxnzr = A->maxnzr;
		{ int taken = 0; LISCHK(try_row_form(A, d, &taken)); if (taken) break; }
		LISCHK(up_i(&d->index, A->index, n * (size_t)A->maxnzr));
		LISCHK(up_d(&d->value, A->value, n * (size_t)A->maxnzr));
		if (!lisg.no_index_codes) {
			int nd = 0;      /* an optimisation: when it cannot be had (out of memory included) the 4 B indices serve */
			int rc = liship_ell_encode_indices(A->n, A->maxnzr, d->index, &d->ell_codes, &d->ell_dict, &nd, lisg.stream);
			if (rc) { (void)liship_free(d->ell_codes); (void)liship_free(d->ell_dict); d->ell_codes = NULL; d->ell_dict = NULL; }
			if (rc && rc != 2 /* hipErrorOutOfMemory */) HIPCHK(rc);
		}
		d->type = LIS_MATRIX_ELL; d->n = A->n;
		fmt_find_plane(d, NULL);
		break;
	case LIS_MATRIX_DIA:
		d->nnd = A->nnd;
		{ int taken = 0; LISCHK(try_row_form(A, d, &taken)); if (taken) break; }
		LISCHK(up_i(&d->index, A->index, (size_t)A->nnd));
		LISCHK(up_d(&d->value, A->value, n * (size_t)A->nnd));
		d->type = LIS_MATRIX_DIA; d->n = A->n;
		fmt_find_plane(d, A->index);
		break;
	case LIS_MATRIX_JAD:
		LISCHK(upload_jad_as_csr(A, d));
		d->type = LIS_MATRIX_CSR;
		LISCHK(lisd_csr_plan(&d->plan, A->n, d->ptr, d->index, d->value));
		break;
	case LIS_MATRIX_BSR:
		d->nr = A->nr; d->nc = A->nc; d->bnr = A->bnr; d->bnc = A->bnc;
		LISCHK(up_i(&d->bptr, A->bptr, (size_t)A->nr + 1));
		LISCHK(up_i(&d->bindex, A->bindex, (size_t)A->bnnz));
		LISCHK(up_d(&d->value, A->value, (size_t)A->bnnz * (size_t)A->bnr * (size_t)A->bnc));
		{	/* constant coefficients: the row form (value records) instead of the native blocks */
			int taken = 0;
			double *native = d->value;
			d->value = NULL;
			LIS_INT e2 = try_bsr_row_form(A->n, A->np, A->bnr, A->bnc, A->is_splited, d, d->bptr, d->bindex, native, A->bnnz, few_distinct_values(A->value, (size_t)A->bnnz * (size_t)A->bnr * (size_t)A->bnc), &taken);
			if (e2) { d->value = native; return e2; }
			if (taken) { (void)liship_free(native); (void)liship_free(d->bptr); (void)liship_free(d->bindex); d->bptr = NULL; d->bindex = NULL; }
			else d->value = native;
		}
		break;
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", A->matrix_type);
	}
	HIPCHK(liship_stream_synchronize(lisg.stream));
	find_inner_rows(A, &d->inner_begin, &d->inner_end);
	d->ready = 1;
	return LIS_SUCCESS;
}

void lisd_mat_free(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	(void)lisp_fill_matrix(A);         /* host arrays still held in HBM only (a matrix converted there) come home before the copy goes */
	if (d->plan) (void)liship_csr_plan_destroy(d->plan);
	if (d->u_plan) (void)liship_csr_plan_destroy(d->u_plan);
	(void)liship_free(d->u_ptr); (void)liship_free(d->u_index); (void)liship_free(d->u_value); (void)liship_free(d->dsplit); (void)liship_free(d->jw);
	if (d->t_plan) (void)liship_csr_plan_destroy(d->t_plan);
	if (d->rt_plan) (void)liship_csr_plan_destroy(d->rt_plan);
	(void)liship_free(d->rt_ptr); (void)liship_free(d->rt_index); (void)liship_free(d->rt_value);
	(void)liship_free(d->t_ptr); (void)liship_free(d->t_index); (void)liship_free(d->t_value); (void)liship_free(d->wr); (void)liship_free(d->t_diag);
	(void)liship_free(d->ell_codes); (void)liship_free(d->ell_dict);
	(void)liship_free(d->ptr); (void)liship_free(d->index); (void)liship_free(d->row);
	(void)liship_free(d->bptr); (void)liship_free(d->bindex); (void)liship_free(d->value);
	(void)liship_free(d->export_index); (void)liship_free(d->ws); free(d->export_run);
	(void)liship_free(d->sx); (void)liship_free(d->sy);
	memset(d, 0, sizeof(*d));
	lisp_matrix_release(A, 0);         /* no copy left that a host write could leave stale: the watched arrays are plain memory again */
}

/* ------------------------------------------------------------------ conversions in HBM (kernels/convert.hip)
 * lis_matrix_convert(Ain, Aout) with Ain = a CSR matrix whose HBM copy exists: the target layout is built FROM that copy by kernels --
 * the arrays the host routines of lis_convert.c build, bit for bit -- and becomes both Aout's HBM copy and the source of Aout's HOST
 * arrays.  Those the Lis API promises (A->index, A->value ...), but a program that only multiplies never reads them: they get address
 * space without access (lis_pages.c), bound to the device buffer that holds their contents, and come home on their first touch.
 * What the product of the new matrix runs on is decided as mat_upload decides it: the constant-coefficient row form (value records)
 * for ELL / DIA where the values allow it, CSC as CSR rows in ascending column order, else the native arrays.
 * *done = 0: not a case for this path (the caller converts on the host).  Single-rank; DIA and CSC want the source rows in ascending
 * column order (csr2dia would sort them in place otherwise -- the host routine does that); rows of more than 96 distinct blocks: host. */
static int device_few_distinct_values(const double *dval, size_t count)
{
	double head[4096];
	const size_t take = count < 4096 ? count : 4096;
	if (take == 0) return 0;
	if (liship_memcpy_d2h(head, dval, take * sizeof(double), lisg.stream) || liship_stream_synchronize(lisg.stream)) return 0;
	return few_distinct_values(head, take);
}

static void *lazy_host(LIS_MATRIX A, size_t bytes, void *dev, int own_dev) { return lisp_alloc_lazy(A, bytes, dev, own_dev); }

LIS_INT lisd_convert_csr(LIS_MATRIX Ain, LIS_MATRIX Aout, int *done)
{
	*done = 0;
	lisd_mat *sd = MDEV(Ain);
	const LIS_INT want = Aout->matrix_type;
	if (want != LIS_MATRIX_ELL && want != LIS_MATRIX_DIA && want != LIS_MATRIX_CSC && want != LIS_MATRIX_BSR && want != LIS_MATRIX_JAD) return LIS_SUCCESS;
	/* (a matrix born in HBM -- lis_amd_matrix_set_csr_device / lis_amd_matrix_poisson3d -- converts like any other: nothing below reads Ain's host arrays, except JAD's
	 * row order, which is the reference's quicksort on the host) */
	if (lisg.no_device_convert || lisg.nprocs > 1 || !lisg.device_ready || (sd->device_only && want == LIS_MATRIX_JAD) ||
	    Ain->matrix_type != LIS_MATRIX_CSR || Ain->is_splited || Ain->np != Ain->n || Ain->n <= 0 || Ain->nnz <= 0)
		return LIS_SUCCESS;
	LISCHK(lisd_mat_ready(Ain));          /* (an upload of the source costs a fraction of a pass of the host routine over it; a stale copy is rebuilt) */
	if (sd->type != LIS_MATRIX_CSR || !sd->ptr || !sd->index || !sd->value) return LIS_SUCCESS;
	const int n = Ain->n, nnz = Ain->nnz;
	lisd_mat *d = MDEV(Aout);
	int *facts = NULL, hfacts[2] = {0, 0};
	HIPCHK(lisd_malloc((void **)&facts, 2 * sizeof(int)));
	int rc = liship_csr_row_facts(n, sd->ptr, sd->index, facts, lisg.stream);
	if (!rc) rc = liship_memcpy_d2h(hfacts, facts, sizeof(hfacts), lisg.stream);
	if (!rc) rc = liship_stream_synchronize(lisg.stream);
	(void)liship_free(facts);
	HIPCHK(rc);
	const int maxlen = hfacts[0], unsorted = hfacts[1];
	LIS_INT err = LIS_SUCCESS;
	memset(d, 0, sizeof(*d));
	d->n = n; d->np = Ain->np; d->nnz = nnz;

	if (want == LIS_MATRIX_ELL) {
		const int maxnzr = maxlen;
		if ((long long)n * maxnzr >= 0x7fffffffLL) return LIS_SUCCESS;
		const size_t slots = (size_t)n * (size_t)maxnzr;
		int *eidx = NULL; double *eval = NULL;
		HIPCHK(lisd_malloc((void **)&eidx, sizeof(int) * (slots ? slots : 1)));
		if (lisd_malloc((void **)&eval, sizeof(double) * (slots ? slots : 1))) { (void)liship_free(eidx); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert\n"); }
		HIPCHK(liship_csr_to_ell(n, maxnzr, sd->ptr, sd->index, sd->value, eidx, eval, lisg.stream));
		int rowform = 0;
		if (!lisg.no_row_form && !lisg.no_value_records && !lisg.no_row_patterns && !lisg.no_index_codes && maxnzr >= 1 && maxnzr <= 32 &&
		    device_few_distinct_values(sd->value, (size_t)nnz)) {
			int *rptr = NULL, *ridx = NULL; double *rval = NULL;
			if (!lisd_malloc((void **)&rptr, sizeof(int) * ((size_t)n + 1)) && !lisd_malloc((void **)&ridx, sizeof(int) * slots) && !lisd_malloc((void **)&rval, sizeof(double) * slots)) {
				HIPCHK(liship_csr_to_ell_rows(n, maxnzr, sd->ptr, sd->index, sd->value, rptr, ridx, rval, lisg.stream));
				d->ptr = rptr; d->index = ridx; d->value = rval;
				err = lisd_csr_plan(&d->plan, n, d->ptr, d->index, d->value);
				if (!err && liship_csr_plan_value_records(d->plan)) { rowform = 1; d->type = LIS_MATRIX_CSR; d->nnz = (int)slots; }
			}
			if (!rowform) {
				if (d->plan) { (void)liship_csr_plan_destroy(d->plan); d->plan = NULL; }
				(void)liship_free(rptr); (void)liship_free(ridx); (void)liship_free(rval);
				d->ptr = NULL; d->index = NULL; d->value = NULL;
				if (err && err != LIS_ERR_OUT_OF_MEMORY) { (void)liship_free(eidx); (void)liship_free(eval); return err; }
				err = LIS_SUCCESS;
			}
		}
		if (!rowform) {
			d->type = LIS_MATRIX_ELL; d->maxnzr = maxnzr; d->index = eidx; d->value = eval;
			if (!lisg.no_index_codes) {
				int nd = 0;
				rc = liship_ell_encode_indices(n, maxnzr, d->index, &d->ell_codes, &d->ell_dict, &nd, lisg.stream);
				if (rc) { (void)liship_free(d->ell_codes); (void)liship_free(d->ell_dict); d->ell_codes = NULL; d->ell_dict = NULL; }
			}
		}
		d->maxnzr = maxnzr;
		if (!rowform) fmt_find_plane(d, NULL);
		LIS_INT *hi = (LIS_INT *)lazy_host(Aout, sizeof(int) * slots, eidx, rowform);
		LIS_SCALAR *hv = (LIS_SCALAR *)lazy_host(Aout, sizeof(double) * slots, eval, rowform);
		if (!hi || !hv) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert: address space\n");
		err = lis_matrix_set_ell(maxnzr, hi, hv, Aout);
	} else if (want == LIS_MATRIX_DIA) {
		if (unsorted) return LIS_SUCCESS;
		const int span = n + Ain->np;
		int *used = NULL, *slot = NULL, *offs = NULL, nnd = 0; long long *scratch = NULL; double *dval = NULL;
		HIPCHK(lisd_malloc((void **)&used, sizeof(int) * (size_t)span));
		if (lisd_malloc((void **)&slot, sizeof(int) * ((size_t)span + 1)) || lisd_malloc((void **)&scratch, sizeof(long long) * ((size_t)span / 4096 + 4))) {
			(void)liship_free(used); (void)liship_free(slot); (void)liship_free(scratch);
			return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert\n");
		}
		rc = liship_csr_dia_offsets(n, Ain->np, sd->ptr, sd->index, used, slot, scratch, &nnd, lisg.stream);
		if (!rc && (nnd <= 0 || (long long)n * nnd >= 0x7fffffffLL)) { (void)liship_free(used); (void)liship_free(slot); (void)liship_free(scratch); return LIS_SUCCESS; }
		if (!rc) rc = lisd_malloc((void **)&offs, sizeof(int) * (size_t)nnd);
		if (!rc) rc = lisd_malloc((void **)&dval, sizeof(double) * (size_t)n * (size_t)nnd);
		if (!rc) rc = liship_csr_to_dia(n, Ain->np, nnd, sd->ptr, sd->index, sd->value, used, slot, offs, dval, lisg.stream);
		(void)liship_stream_synchronize(lisg.stream);
		(void)liship_free(used); (void)liship_free(slot);
		if (rc) { (void)liship_free(scratch); (void)liship_free(offs); (void)liship_free(dval); HIPCHK(rc); }
		int rowform = 0;
		if (!lisg.no_row_form && !lisg.no_value_records && !lisg.no_row_patterns && !lisg.no_index_codes && nnd <= 32 &&
		    device_few_distinct_values(sd->value, (size_t)nnz)) {
			int *count = NULL, *rptr = NULL, *ridx = NULL, rnnz = 0; double *rval = NULL;
			if (!lisd_malloc((void **)&count, sizeof(int) * (size_t)n) && !lisd_malloc((void **)&rptr, sizeof(int) * ((size_t)n + 1)) &&
			    !liship_dia_row_counts(n, Ain->np, nnd, offs, count, rptr, scratch, &rnnz, lisg.stream) && rnnz > 0 &&
			    !lisd_malloc((void **)&ridx, sizeof(int) * (size_t)rnnz) && !lisd_malloc((void **)&rval, sizeof(double) * (size_t)rnnz) &&
			    !liship_dia_to_rows(n, Ain->np, nnd, offs, dval, rptr, ridx, rval, lisg.stream)) {
				d->ptr = rptr; d->index = ridx; d->value = rval;
				err = lisd_csr_plan(&d->plan, n, d->ptr, d->index, d->value);
				if (!err && liship_csr_plan_value_records(d->plan)) { rowform = 1; d->type = LIS_MATRIX_CSR; d->nnz = rnnz; }
			}
			(void)liship_free(count);
			if (!rowform) {
				if (d->plan) { (void)liship_csr_plan_destroy(d->plan); d->plan = NULL; }
				(void)liship_free(rptr); (void)liship_free(ridx); (void)liship_free(rval);
				d->ptr = NULL; d->index = NULL; d->value = NULL;
				if (err && err != LIS_ERR_OUT_OF_MEMORY) { (void)liship_free(scratch); (void)liship_free(offs); (void)liship_free(dval); return err; }
				err = LIS_SUCCESS;
			}
		}
		(void)liship_free(scratch);
		if (!rowform) { d->type = LIS_MATRIX_DIA; d->index = offs; d->value = dval; }
		d->nnd = nnd;
		if (!rowform && nnd <= 4096) {
			int hoffs[4096];
			if (liship_memcpy_d2h(hoffs, offs, sizeof(int) * (size_t)nnd, lisg.stream) == 0 && liship_stream_synchronize(lisg.stream) == 0) fmt_find_plane(d, hoffs);
		}
		LIS_INT *hi = (LIS_INT *)lazy_host(Aout, sizeof(int) * (size_t)nnd, offs, rowform);
		LIS_SCALAR *hv = (LIS_SCALAR *)lazy_host(Aout, sizeof(double) * (size_t)n * (size_t)nnd, dval, rowform);
		if (!hi || !hv) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert: address space\n");
		err = lis_matrix_set_dia(nnd, hi, hv, Aout);
	} else if (want == LIS_MATRIX_CSC) {
		if (unsorted) return LIS_SUCCESS;
		/* the host arrays: A^T row by row = A column by column, rows ascending inside a column (transpose.hip); the product's arrays: A's
		 * own rows, which ARE in ascending column order here -- the order the reference's serial CSC sweep adds them in (lis_matvec_csc.c:128-144) */
		int *tptr = NULL, *tidx = NULL; double *tval = NULL; void *work = NULL;
		const size_t wbytes = sizeof(int) * ((size_t)Ain->np + (size_t)nnz) + 16;      /* (as lis_matvech.c sizes it) */
		if (lisd_malloc((void **)&tptr, sizeof(int) * ((size_t)Ain->np + 1)) || lisd_malloc((void **)&tidx, sizeof(int) * (size_t)nnz) ||
		    lisd_malloc((void **)&tval, sizeof(double) * (size_t)nnz) || lisd_malloc(&work, wbytes)) {
			(void)liship_free(tptr); (void)liship_free(tidx); (void)liship_free(tval); (void)liship_free(work);
			return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert\n");
		}
		rc = liship_csr_transpose_f64(n, Ain->np, nnz, sd->ptr, sd->index, sd->value, tptr, tidx, tval, work, lisg.stream);
		if (!rc) rc = liship_stream_synchronize(lisg.stream);
		(void)liship_free(work);
		if (!rc) rc = lisd_malloc((void **)&d->ptr, sizeof(int) * ((size_t)n + 1));
		if (!rc) rc = lisd_malloc((void **)&d->index, sizeof(int) * (size_t)nnz);
		if (!rc) rc = lisd_malloc((void **)&d->value, sizeof(double) * (size_t)nnz);
		if (!rc) rc = liship_memcpy_d2d(d->ptr, sd->ptr, sizeof(int) * ((size_t)n + 1), lisg.stream);
		if (!rc) rc = liship_memcpy_d2d(d->index, sd->index, sizeof(int) * (size_t)nnz, lisg.stream);
		if (!rc) rc = liship_memcpy_d2d(d->value, sd->value, sizeof(double) * (size_t)nnz, lisg.stream);
		if (rc) { (void)liship_free(tptr); (void)liship_free(tidx); (void)liship_free(tval); HIPCHK(rc); }
		d->type = LIS_MATRIX_CSR;
		err = lisd_csr_plan(&d->plan, n, d->ptr, d->index, d->value);
		if (err) { (void)liship_free(tptr); (void)liship_free(tidx); (void)liship_free(tval); return err; }
		LIS_INT *hp = (LIS_INT *)lazy_host(Aout, sizeof(int) * ((size_t)Ain->np + 1), tptr, 1);
		LIS_INT *hi = (LIS_INT *)lazy_host(Aout, sizeof(int) * (size_t)nnz, tidx, 1);
		LIS_SCALAR *hv = (LIS_SCALAR *)lazy_host(Aout, sizeof(double) * (size_t)nnz, tval, 1);
		if (!hp || !hi || !hv) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert: address space\n");
		err = lis_matrix_set_csc(nnz, hp, hi, hv, Aout);
	} else if (want == LIS_MATRIX_JAD) {
		/* the row order is the reference's unstable quicksort of the row lengths: made on the host (its recursion spread over the threads,
		 * lis_convert.c), like the diagonal starts; the entries are placed here.  The product's arrays are A's own: the j-th entry of a
		 * row sits on jagged diagonal j, so rows in their original order, entries in theirs, is the order lis_matvec_jad adds them in */
		LIS_INT maxnzr = 0, *perm = NULL, *jptr = NULL;
		LISCHK(lisi_jad_order(Ain, &maxnzr, &perm, &jptr));
		int *dperm = NULL, *djptr = NULL, *jidx = NULL; double *jval = NULL;
		rc = lisd_malloc((void **)&dperm, sizeof(int) * (size_t)n);
		if (!rc) rc = lisd_malloc((void **)&djptr, sizeof(int) * ((size_t)maxnzr + 1));
		if (!rc) rc = lisd_malloc((void **)&jidx, sizeof(int) * (size_t)nnz);
		if (!rc) rc = lisd_malloc((void **)&jval, sizeof(double) * (size_t)nnz);
		if (!rc) rc = liship_memcpy_h2d(dperm, perm, sizeof(int) * (size_t)n, lisg.stream);
		if (!rc) rc = liship_memcpy_h2d(djptr, jptr, sizeof(int) * ((size_t)maxnzr + 1), lisg.stream);
		if (!rc) rc = liship_csr_to_jad(n, dperm, djptr, sd->ptr, sd->index, sd->value, jidx, jval, lisg.stream);
		if (!rc) rc = lisd_malloc((void **)&d->ptr, sizeof(int) * ((size_t)n + 1));
		if (!rc) rc = lisd_malloc((void **)&d->index, sizeof(int) * (size_t)nnz);
		if (!rc) rc = lisd_malloc((void **)&d->value, sizeof(double) * (size_t)nnz);
		if (!rc) rc = liship_memcpy_d2d(d->ptr, sd->ptr, sizeof(int) * ((size_t)n + 1), lisg.stream);
		if (!rc) rc = liship_memcpy_d2d(d->index, sd->index, sizeof(int) * (size_t)nnz, lisg.stream);
		if (!rc) rc = liship_memcpy_d2d(d->value, sd->value, sizeof(double) * (size_t)nnz, lisg.stream);
		if (!rc) rc = liship_stream_synchronize(lisg.stream);
		(void)liship_free(dperm); (void)liship_free(djptr);
		if (rc) { free(perm); free(jptr); (void)liship_free(jidx); (void)liship_free(jval); HIPCHK(rc); }
		d->type = LIS_MATRIX_CSR;
		err = lisd_csr_plan(&d->plan, n, d->ptr, d->index, d->value);
		if (err) { free(perm); free(jptr); (void)liship_free(jidx); (void)liship_free(jval); return err; }
		LIS_INT *hi = (LIS_INT *)lazy_host(Aout, sizeof(int) * (size_t)nnz, jidx, 1);
		LIS_SCALAR *hv = (LIS_SCALAR *)lazy_host(Aout, sizeof(double) * (size_t)nnz, jval, 1);
		if (!hi || !hv) { free(perm); free(jptr); return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert: address space\n"); }
		err = lis_matrix_set_jad(nnz, maxnzr, perm, jptr, hi, hv, Aout);
	} else {                                              /* BSR */
		const int bnr = Aout->conv_bnr, bnc = Aout->conv_bnc;
		if (bnr < 1 || bnc < 1) return LIS_SUCCESS;
		const int nr = 1 + (n - 1) / bnr, pad = (bnc - n % bnc) % bnc;
		int *count = NULL, *bptr = NULL, *bindex = NULL, bnnz = 0; long long *scratch = NULL; double *bval = NULL;
		if (lisd_malloc((void **)&count, sizeof(int) * ((size_t)nr + 1)) || lisd_malloc((void **)&bptr, sizeof(int) * ((size_t)nr + 1)) ||
		    lisd_malloc((void **)&scratch, sizeof(long long) * ((size_t)nr / 4096 + 4))) {
			(void)liship_free(count); (void)liship_free(bptr); (void)liship_free(scratch);
			return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert\n");
		}
		rc = liship_csr_bsr_count(n, Ain->np, bnr, bnc, sd->ptr, sd->index, count, bptr, scratch, &bnnz, lisg.stream);
		(void)liship_free(count); (void)liship_free(scratch);
		if (rc || bnnz <= 0 || (long long)bnnz * bnr * bnc >= 0x7fffffffLL) { (void)liship_free(bptr); if (rc) HIPCHK(rc); return LIS_SUCCESS; }
		if (!rc) rc = lisd_malloc((void **)&bindex, sizeof(int) * (size_t)bnnz);
		if (!rc) rc = lisd_malloc((void **)&bval, sizeof(double) * (size_t)bnnz * (size_t)bnr * (size_t)bnc);
		if (!rc) rc = liship_csr_to_bsr(n, bnr, bnc, bnnz, sd->ptr, sd->index, sd->value, bptr, bindex, bval, lisg.stream);
		if (rc) { (void)liship_free(bptr); (void)liship_free(bindex); (void)liship_free(bval); HIPCHK(rc); }
		d->type = LIS_MATRIX_BSR; d->nr = nr; d->bnr = bnr; d->bnc = bnc;
		int rowform = 0;
		if (pad == 0) {          /* constant coefficients: the row form (try_bsr_row_form); Aout's header is not filled in yet, the source's facts are the target's */
			err = try_bsr_row_form(n, Ain->np, bnr, bnc, 0, d, bptr, bindex, bval, bnnz, device_few_distinct_values(sd->value, (size_t)nnz), &rowform);
			if (err) { (void)liship_free(bptr); (void)liship_free(bindex); (void)liship_free(bval); return err; }
		}
		if (!rowform) { d->bptr = bptr; d->bindex = bindex; d->value = bval; }
		/* (with the row form the native arrays only back the host arrays: they go when those have been read or the matrix dies) */
		LIS_INT *hp = (LIS_INT *)lazy_host(Aout, sizeof(int) * ((size_t)nr + 1), bptr, rowform);
		LIS_INT *hi = (LIS_INT *)lazy_host(Aout, sizeof(int) * (size_t)bnnz, bindex, rowform);
		LIS_SCALAR *hv = (LIS_SCALAR *)lazy_host(Aout, sizeof(double) * (size_t)bnnz * (size_t)bnr * (size_t)bnc, bval, rowform);
		if (!hp || !hi || !hv) {
			/* the pages that were made go (with the row form they own their buffer and free it), a buffer whose pages were not made is freed here; without the
			 * row form the buffers are d's and go with the half-made copy (lisd_mat_free in the caller's error path) */
			if (hp) (void)lisp_free_array(hp); else if (rowform) (void)liship_free(bptr);
			if (hi) (void)lisp_free_array(hi); else if (rowform) (void)liship_free(bindex);
			if (hv) (void)lisp_free_array(hv); else if (rowform) (void)liship_free(bval);
			return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "convert: address space\n");
		}
		err = lis_matrix_set_bsr(bnr, bnc, bnnz, hp, hi, hv, Aout);
		if (!err) { Aout->pad_comm = pad; d->nc = Aout->nc; }
	}
	if (err) return err;
	HIPCHK(liship_stream_synchronize(lisg.stream));
	d->inner_begin = 0; d->inner_end = n;
	d->ready = 1;                                      /* the HBM copy exists: lis_matrix_assemble's eager upload finds nothing to do */
	err = lis_matrix_assemble(Aout);
	if (err) { lisi_matrix_storage_destroy(Aout); return err; }
	*done = 1;
	return LIS_SUCCESS;
}

LIS_INT lis_amd_matrix_upload(LIS_MATRIX A) { return lisd_mat_ready(A); }
LIS_INT lis_amd_matrix_index_codes(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_coded(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_row_patterns(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_row_patterns(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_pattern_records(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_pattern_records(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_value_records(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_value_records(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_dominant_pattern(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_dominant_pattern(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_wide_dominant(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_wide_dominant(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_marching(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	if (!(MDEV(A)->type == LIS_MATRIX_CSR && MDEV(A)->plan)) return 0;
	if (liship_csr_plan_block2_march(MDEV(A)->plan)) return 4;
	if (liship_csr_plan_box27(MDEV(A)->plan)) return 3;
	return liship_csr_plan_marching(MDEV(A)->plan);
}
LIS_INT lis_amd_matrix_strip_rows(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_strip_rows(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_block_rows(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_block_rows(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_matrix_device_type(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->type;
}
LIS_INT lis_amd_matrix_local_columns(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	const long long listed = MDEV(A)->plan ? liship_csr_plan_localized(MDEV(A)->plan) : 0;
	return listed > 0x7fffffffLL ? 0x7fffffff : (LIS_INT)listed;
}
void *lis_amd_matrix_csr_plan(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return NULL;
	return MDEV(A)->type == LIS_MATRIX_CSR ? (void *)MDEV(A)->plan : NULL;
}
long long lis_amd_matrix_reordered(LIS_MATRIX A)
{
	if (lisd_mat_ready(A) != LIS_SUCCESS) return 0;
	return MDEV(A)->plan ? liship_csr_plan_reordered(MDEV(A)->plan) : 0;
}
LIS_INT lis_amd_set_matrix_check(LIS_INT on) { lisg.matrix_check = on ? 1 : 0; return LIS_SUCCESS; }
LIS_INT lis_amd_matrix_host_written(LIS_MATRIX A) { return MDEV(A)->host_written; }
/* tests of the write watch without a GPU: the arrays of an assembled matrix are adopted and write-protected exactly as lisd_mat_ready does after an upload */
LIS_INT lis_amd_matrix_page_test_watch(LIS_MATRIX A)
{
	const void *arr[6]; size_t bytes[6];
	const int k = host_arrays(A, arr, bytes);
	MDEV(A)->host_written = 0;
	for (int i = 0; i < k; i++) (void)lisp_adopt(A, (void *)arr[i]);
	return lisp_matrix_protect(A);
}
LIS_INT lis_amd_matrix_host_modified(LIS_MATRIX A)
{
	if (MDEV(A)->device_only) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix lives in HBM only\n");
	lisd_mat_free(A);
	return LIS_SUCCESS;
}

/* y = A x on device pointers.  In a multi-GPU job the ghost part of x is filled first; the rows that do
 * not touch ghosts are issued before the exchange completes (same stream ordering via RCCL). */
LIS_INT lisd_spmv(LIS_MATRIX A, double *dx, double *dy)
{
	lisd_mat *d = MDEV(A);
	LISCHK(lisd_mat_ready(A));
	d->served++;
	if (lisg.nprocs > 1 && A->commtable && d->type == LIS_MATRIX_CSR && !lisg.no_overlap &&
	    d->inner_end - d->inner_begin >= d->n / 2) {
		/* rows [inner_begin, inner_end) reference no ghost column: they run while the halo is in flight; the
		 * boundary rows follow once the ghosts have landed (same kernel, same bits: rows are independent) */
		LISCHK(lisc_halo_begin(A, dx));
		HIPCHK(liship_spmv_csr_rows_f64(d->plan, d->inner_begin, d->inner_end, d->ptr, d->index, d->value, dx, dy, lisg.stream));
		LISCHK(lisc_halo_end(A, dx));
		if (d->inner_begin > 0)
			HIPCHK(liship_spmv_csr_rows_f64(d->plan, 0, d->inner_begin, d->ptr, d->index, d->value, dx, dy, lisg.stream));
		if (d->inner_end < d->n)
			HIPCHK(liship_spmv_csr_rows_f64(d->plan, d->inner_end, d->n, d->ptr, d->index, d->value, dx, dy, lisg.stream));
		return LIS_SUCCESS;
	}
	if (lisg.nprocs > 1 && A->commtable && (d->type == LIS_MATRIX_ELL || d->type == LIS_MATRIX_DIA) && !lisg.no_overlap &&
	    d->inner_end - d->inner_begin >= d->n / 2) {
		/* the native ELL / DIA layouts likewise: interior rows under the halo, boundary rows behind it (an odd cut makes that part run
		 * one row per lane) */
		LISCHK(lisc_halo_begin(A, dx));
		for (int part = 0; part < 3; part++) {
			const int rb = part == 0 ? d->inner_begin : part == 1 ? 0 : d->inner_end, re = part == 0 ? d->inner_end : part == 1 ? d->inner_begin : d->n;
			if (part == 1) LISCHK(lisc_halo_end(A, dx));
			if (rb >= re) continue;
			if (d->type == LIS_MATRIX_ELL) HIPCHK(liship_spmv_ell_rows_f64(d->n, d->maxnzr, d->index, d->ell_codes, d->ell_dict, d->value, dx, dy, rb, re, lisg.stream));
			else HIPCHK(liship_spmv_dia_rows_f64(d->n, d->np, d->nnd, d->index, d->value, dx, dy, rb, re, lisg.stream));
		}
		return LIS_SUCCESS;
	}
	if (lisg.nprocs > 1 && A->commtable && d->type == LIS_MATRIX_BSR && !lisg.no_overlap && d->inner_end - d->inner_begin >= d->nr / 2) {
		/* BSR likewise, in block rows (inner_begin / inner_end count block rows for this format) */
		LISCHK(lisc_halo_begin(A, dx));
		HIPCHK(liship_spmv_bsr_rows_f64(d->nr, A->bnnz, d->bnr, d->bnc, d->bptr, d->bindex, d->value, dx, dy, d->inner_begin, d->inner_end, lisg.stream));
		LISCHK(lisc_halo_end(A, dx));
		HIPCHK(liship_spmv_bsr_rows_f64(d->nr, A->bnnz, d->bnr, d->bnc, d->bptr, d->bindex, d->value, dx, dy, 0, d->inner_begin, lisg.stream));
		HIPCHK(liship_spmv_bsr_rows_f64(d->nr, A->bnnz, d->bnr, d->bnc, d->bptr, d->bindex, d->value, dx, dy, d->inner_end, d->nr, lisg.stream));
		return LIS_SUCCESS;
	}
	if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dx));
	if (d->split_jad) {
		/* y = (D x + L x) + U x, the two sparse sums each formed from 0 on their own: w = L x; w = D.*x + 1*w (exact: 1*w is w);
		 * y = U x; y += 1*w (a + b and b + a are the same double) */
		HIPCHK(liship_spmv_csr_f64(d->plan, d->ptr, d->index, d->value, dx, d->jw, lisg.stream));
		HIPCHK(liship_pmul_xpay_f64(d->n, d->dsplit, dx, 1.0, d->jw, lisg.stream));
		HIPCHK(liship_spmv_csr_f64(d->u_plan, d->u_ptr, d->u_index, d->u_value, dx, dy, lisg.stream));
		HIPCHK(liship_axpy_f64(d->n, 1.0, d->jw, dy, lisg.stream));
		return LIS_SUCCESS;
	}
	switch (d->type) {
	case LIS_MATRIX_CSR:
		HIPCHK(liship_spmv_csr_f64(d->plan, d->ptr, d->index, d->value, dx, dy, lisg.stream));
		break;
	case LIS_MATRIX_ELL:
		fmt_strips(d);
		if (d->ell_codes) {
			int rc = liship_spmv_ell_coded_f64(d->n, d->maxnzr, d->ell_codes, d->ell_dict, d->value, dx, dy, NULL, -1, NULL, NULL, lisg.stream);
			if (rc == 0) break;
			if (rc != LISHIP_ERR_ARG) HIPCHK(rc);
		}
		HIPCHK(liship_spmv_ell_f64(d->n, d->maxnzr, d->index, d->value, dx, dy, lisg.stream));
		break;
	case LIS_MATRIX_DIA:
		fmt_strips(d);
		HIPCHK(liship_spmv_dia_f64(d->n, d->np, d->nnd, d->index, d->value, dx, dy, lisg.stream));
		break;
	case LIS_MATRIX_JAD:
		HIPCHK(liship_spmv_jad_f64(d->n, d->maxnzr, d->row, d->ptr, d->index, d->value, dx, dy, lisg.stream));
		break;
	case LIS_MATRIX_BSR:
		HIPCHK(liship_spmv_bsr_nnz_f64(d->nr, A->bnnz, d->bnr, d->bnc, d->bptr, d->bindex, d->value, dx, dy, lisg.stream));
		break;
	default:
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "storage format %D is not served by liblis_amd\n", d->type);
	}
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ reductions -> host scalars */
/* y = A x with <w,y> (and <y,y>) formed in the product's epilogue when the kernel can (CSR row-gather);
 * otherwise the product followed by one reduction pass.  The sums land in lisg.reduce_out[0..1]. */
LIS_INT lisd_spmv_dot_launch(LIS_MATRIX A, double *dx, double *dy, const double *dw, int want_sumsq)
{
	return lisd_spmv_dot_launch_to(A, dx, dy, dw, want_sumsq, lisg.reduce_out);
}

LIS_INT lisd_spmv_dot_launch_to(LIS_MATRIX A, double *dx, double *dy, const double *dw, int want_sumsq, double *result)
{
	lisd_mat *d = MDEV(A);
	LISCHK(lisd_mat_ready(A));
	d->served++;                          /* (a branch below that falls back to lisd_spmv counts the product twice: the count is a threshold, not a statistic) */
	int nblocks = 0;
	if (d->split_jad) {
		LISCHK(lisd_spmv(A, dx, dy));
		if (want_sumsq) HIPCHK(liship_dot2_f64(d->n, dy, dw, result, lisg.reduce_work, lisg.stream));
		else HIPCHK(liship_dot_f64(d->n, dw, dy, result, lisg.reduce_work, lisg.stream));
		return LIS_SUCCESS;
	}
	if (d->type == LIS_MATRIX_CSR && d->plan) (void)liship_csr_plan_info(d->plan, NULL, NULL, &nblocks);
	/* the three parts launch at most nblocks + 2 row blocks (each cut splits one): all of them must find a slot for
	 * their partial sums BEFORE the first part is launched -- otherwise the plain overlapped product + one dot pass */
	const int slots_ok = (d->type == LIS_MATRIX_CSR && d->plan)
		? (size_t)liship_csr_plan_fused_slots(d->plan) <= liship_reduce_work_bytes() / sizeof(double) / 4
		: (size_t)nblocks + 2 <= liship_reduce_work_bytes() / sizeof(double) / 4;
	/* a plan whose products run the team / staged kernels has no per-row-block epilogue: the plain (overlapped) product and one reduction pass */
	const int fused = !(d->type == LIS_MATRIX_CSR && d->plan) || liship_csr_plan_fused_dots(d->plan);
	if (d->type == LIS_MATRIX_CSR && !fused) {
		LISCHK(lisd_spmv(A, dx, dy));
	} else if (d->type == LIS_MATRIX_CSR && !lisg.no_fusion && lisg.nprocs > 1 && A->commtable && !lisg.no_overlap &&
	    d->inner_end - d->inner_begin >= d->n / 2 && !slots_ok) {
		LISCHK(lisd_spmv(A, dx, dy));
	} else if (d->type == LIS_MATRIX_CSR && !lisg.no_fusion && lisg.nprocs > 1 && A->commtable && !lisg.no_overlap &&
	    d->inner_end - d->inner_begin >= d->n / 2) {
		/* as lisd_spmv: interior rows while the halo travels, boundary rows after it; every part parks its
		 * per-block partial sums, one fold at the end */
		int used = 0, total = 0;
		LISCHK(lisc_halo_begin(A, dx));
		int rc = liship_spmv_csr_rows_dot_f64(d->plan, d->inner_begin, d->inner_end, d->ptr, d->index, d->value, dx, dy, dw,
		                                      want_sumsq, lisg.reduce_work, 0, &used, lisg.stream);
		LISCHK(lisc_halo_end(A, dx));
		if (rc == 0) {
			total = used;
			if (d->inner_begin > 0) {
				HIPCHK(liship_spmv_csr_rows_dot_f64(d->plan, 0, d->inner_begin, d->ptr, d->index, d->value, dx, dy, dw,
				                                    want_sumsq, lisg.reduce_work, total, &used, lisg.stream));
				total += used;
			}
			if (d->inner_end < d->n) {
				HIPCHK(liship_spmv_csr_rows_dot_f64(d->plan, d->inner_end, d->n, d->ptr, d->index, d->value, dx, dy, dw,
				                                    want_sumsq, lisg.reduce_work, total, &used, lisg.stream));
				total += used;
			}
			HIPCHK(liship_spmv_csr_dot_finish_f64(total, want_sumsq, result, lisg.reduce_work, lisg.stream));
			return LIS_SUCCESS;
		}
		if (rc != LISHIP_ERR_ARG) HIPCHK(rc);
		HIPCHK(liship_spmv_csr_f64(d->plan, d->ptr, d->index, d->value, dx, dy, lisg.stream));   /* the ghosts are in */
	} else if (d->type == LIS_MATRIX_CSR && !lisg.no_fusion) {
		if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dx));
		int rc = liship_spmv_csr_dot_f64(d->plan, d->ptr, d->index, d->value, dx, dy, dw, want_sumsq,
		                                 result, lisg.reduce_work, lisg.stream);
		if (rc == 0) return LIS_SUCCESS;
		if (rc != LISHIP_ERR_ARG) HIPCHK(rc);
		HIPCHK(liship_spmv_csr_f64(d->plan, d->ptr, d->index, d->value, dx, dy, lisg.stream));
	} else if ((d->type == LIS_MATRIX_ELL || d->type == LIS_MATRIX_DIA) && !lisg.no_fusion) {
		if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dx));
		fmt_strips(d);
		int rc = (d->type == LIS_MATRIX_ELL && d->ell_codes)
			? liship_spmv_ell_coded_f64(d->n, d->maxnzr, d->ell_codes, d->ell_dict, d->value, dx, dy, dw, want_sumsq ? 1 : 0, result, lisg.reduce_work, lisg.stream)
			: (d->type == LIS_MATRIX_ELL)
			? liship_spmv_ell_dot_f64(d->n, d->maxnzr, d->index, d->value, dx, dy, dw, want_sumsq, result, lisg.reduce_work, lisg.stream)
			: liship_spmv_dia_dot_f64(d->n, d->np, d->nnd, d->index, d->value, dx, dy, dw, want_sumsq, result, lisg.reduce_work, lisg.stream);
		if (rc == 0) return LIS_SUCCESS;
		if (rc != LISHIP_ERR_ARG) HIPCHK(rc);
		if (d->type == LIS_MATRIX_ELL) HIPCHK(liship_spmv_ell_f64(d->n, d->maxnzr, d->index, d->value, dx, dy, lisg.stream));
		else HIPCHK(liship_spmv_dia_f64(d->n, d->np, d->nnd, d->index, d->value, dx, dy, lisg.stream));
	} else if (d->type == LIS_MATRIX_BSR && d->bnr == d->bnc && !lisg.no_fusion) {
		if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dx));
		int rc = liship_spmv_bsr_dot_f64(d->nr, d->n, A->bnnz, d->bnr, d->bptr, d->bindex, d->value, dx, dy, dw, want_sumsq,
		                                 result, lisg.reduce_work, lisg.stream);
		if (rc == 0) return LIS_SUCCESS;
		if (rc != LISHIP_ERR_ARG) HIPCHK(rc);
		HIPCHK(liship_spmv_bsr_nnz_f64(d->nr, A->bnnz, d->bnr, d->bnc, d->bptr, d->bindex, d->value, dx, dy, lisg.stream));
	} else LISCHK(lisd_spmv(A, dx, dy));
	if (want_sumsq) HIPCHK(liship_dot2_f64(d->n, dy, dw, result, lisg.reduce_work, lisg.stream));
	else HIPCHK(liship_dot_f64(d->n, dw, dy, result, lisg.reduce_work, lisg.stream));
	return LIS_SUCCESS;
}

LIS_INT lisd_fetch(int count, double *out)
{
	if (lisg.nprocs > 1 && lisg.comm_kind == 1)        /* RCCL gathers the partials straight from HBM: one sync, not two */
		return lisc_fold(count, out);
	HIPCHK(liship_memcpy_d2h(lisg.host_out, lisg.reduce_out, (size_t)count * sizeof(double), lisg.stream));
	HIPCHK(liship_stream_synchronize(lisg.stream));
	for (int i = 0; i < count; i++) out[i] = lisg.host_out[i];
	if (lisg.nprocs > 1) LISCHK(lisc_fold(count, out));
	return LIS_SUCCESS;
}

LIS_INT lisd_dot(int n, const double *dx, const double *dy, double *out)
{
	HIPCHK(liship_dot_f64(n, dx, dy, lisg.reduce_out, lisg.reduce_work, lisg.stream));
	return lisd_fetch(1, out);
}

LIS_INT lisd_nrm2(int n, const double *dx, double *out)
{
	/* the root is taken after the cross-rank fold (ref lis_vector_ops.c:263-264) */
	HIPCHK(liship_sumsq_f64(n, dx, lisg.reduce_out, lisg.reduce_work, lisg.stream));
	LISCHK(lisd_fetch(1, out));
	*out = sqrt(*out);
	return LIS_SUCCESS;
}

LIS_INT lisd_nrm1(int n, const double *dx, double *out)
{
	HIPCHK(liship_nrm1_f64(n, dx, lisg.reduce_out, lisg.reduce_work, lisg.stream));
	return lisd_fetch(1, out);
}

LIS_INT lisd_dot2(int n, const double *dx, const double *dy, double *out2)
{
	HIPCHK(liship_dot2_f64(n, dx, dy, lisg.reduce_out, lisg.reduce_work, lisg.stream));
	return lisd_fetch(2, out2);
}

/* ------------------------------------------------------------------ matrices born in HBM */
LIS_INT lis_amd_matrix_set_csr_device(LIS_INT nnz, LIS_INT np, LIS_INT *dptr, LIS_INT *dindex,
                                      LIS_SCALAR *dvalue, LIS_MATRIX A)
{
	if (!lisi_is_registered(A)) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is undefined\n");
	if (A->status != LIS_MATRIX_NULL) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A must be sized and not yet assembled\n");
	if (np < A->n) return LISI_ERR(LIS_ERR_ILL_ARG, "np(=%D) is smaller than n(=%D)\n", np, A->n);
	LISCHK(lisd_init());
	lisd_mat *d = MDEV(A);
	d->device_only = 1;
	d->type = LIS_MATRIX_CSR;
	d->n = A->n; d->np = np; d->nnz = nnz;
	d->ptr = dptr; d->index = dindex; d->value = dvalue;
	LISCHK(lisd_csr_plan_cols(&d->plan, A->n, np, d->ptr, d->index, d->value));
	d->inner_begin = 0; d->inner_end = A->n;
	d->ready = 1;
	A->nnz = nnz; A->np = np;
	A->is_copy = LIS_FALSE;
	A->matrix_type = LIS_MATRIX_CSR;
	A->status = LIS_MATRIX_CSR;                /* assembled: there is nothing left to do on the host */
	return LIS_SUCCESS;
}

LIS_INT lis_amd_matrix_poisson3d(LIS_MATRIX A, LIS_INT l, LIS_INT m, LIS_INT n, LIS_INT sorted)
{
	if (!lisi_is_registered(A)) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A is undefined\n");
	if (A->status != LIS_MATRIX_NULL) return LISI_ERR(LIS_ERR_ILL_ARG, "matrix A must be sized and not yet assembled\n");
	const long long mn = (long long)m * n, gn = mn * l;
	if (gn != A->gn) return LISI_ERR(LIS_ERR_ILL_ARG, "grid %D x %D x %D does not match the global size\n", l, m, n);
	const long long nnz = liship_poisson3d_nnz(l, m, n, A->is, A->ie);
	if (nnz < 0) return LISI_ERR(LIS_ERR_ILL_ARG, "rows [%D,%D) are not whole grid planes of %D rows\n", A->is, A->ie, (LIS_INT)mn);
	LISCHK(lisd_init());
	const LIS_INT nloc = A->n, nlow = A->is > 0 ? (LIS_INT)mn : 0, nup = A->ie < gn ? (LIS_INT)mn : 0;
	int *dptr = NULL, *didx = NULL; double *dval = NULL;
	HIPCHK(lisd_malloc((void **)&dptr, sizeof(int) * ((size_t)nloc + 1 + 4)));
	HIPCHK(lisd_malloc((void **)&didx, sizeof(int) * ((size_t)nnz + 4)));
	HIPCHK(lisd_malloc((void **)&dval, sizeof(double) * ((size_t)nnz + 2)));
	HIPCHK(liship_poisson3d_csr(l, m, n, A->is, A->ie, sorted, dptr, didx, dval, lisg.stream));
	LISCHK(lis_amd_matrix_set_csr_device((LIS_INT)nnz, nloc + nlow + nup, dptr, didx, dval, A));
	if (A->nprocs > 1) {
		/* halo tables in closed form: one plane to/from the rank below and above (ghosts ascending: low first) */
		const LIS_INT ns = nlow + nup;
		A->l2g_map = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(ns > 0 ? ns : 1));
		for (LIS_INT i = 0; i < nlow; i++) A->l2g_map[i] = A->is - (LIS_INT)mn + i;
		for (LIS_INT i = 0; i < nup; i++) A->l2g_map[nlow + i] = A->ie + i;
		LIS_COMMTABLE t = (LIS_COMMTABLE)calloc(1, sizeof(struct LIS_COMMTABLE_STRUCT));
		const LIS_INT nb = (nlow ? 1 : 0) + (nup ? 1 : 0);
		t->neibpe = (LIS_INT *)malloc(sizeof(LIS_INT) * 2);
		t->import_ptr = (LIS_INT *)calloc(3, sizeof(LIS_INT)); t->export_ptr = (LIS_INT *)calloc(3, sizeof(LIS_INT));
		t->import_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(ns > 0 ? ns : 1));
		t->export_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(ns > 0 ? ns : 1));
		LIS_INT q = 0;
		if (nlow) { t->neibpe[q] = A->my_rank - 1; t->import_ptr[q + 1] = t->import_ptr[q] + nlow; t->export_ptr[q + 1] = t->export_ptr[q] + nlow;
			for (LIS_INT i = 0; i < nlow; i++) t->export_index[t->export_ptr[q] + i] = i; q++; }
		if (nup)  { t->neibpe[q] = A->my_rank + 1; t->import_ptr[q + 1] = t->import_ptr[q] + nup;  t->export_ptr[q + 1] = t->export_ptr[q] + nup;
			for (LIS_INT i = 0; i < nup; i++) t->export_index[t->export_ptr[q] + i] = nloc - (LIS_INT)mn + i; q++; }
		for (LIS_INT i = 0; i < ns; i++) t->import_index[i] = nloc + i;
		t->comm = A->comm; t->neibpetot = nb; t->imnnz = ns; t->exnnz = ns; t->wssize = ns; t->wrsize = ns;
		t->ws = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(ns > 0 ? ns : 1));
		t->wr = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(ns > 0 ? ns : 1));
		A->commtable = t;
		A->is_comm = LIS_TRUE;
		MDEV(A)->inner_begin = nlow; MDEV(A)->inner_end = nloc - nup;
	}
	return LIS_SUCCESS;
}

LIS_INT lis_amd_vector_poisson3d_rhs(LIS_VECTOR b, LIS_INT l, LIS_INT m, LIS_INT n)
{
	double *db;
	LISCHK(lisd_vec_out(b, &db));
	HIPCHK(liship_poisson3d_rhs(l, m, n, b->is, b->ie, db, lisg.stream));
	return lisd_vec_done(b);
}
