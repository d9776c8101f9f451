/*
 * lis_pages.c -- coherent semantics at resident speed: the host pages of a vector follow its HBM copy through page protection.
 *
 * Lis hands out raw host arrays (v->value[], lis.h:513-537) and programs read and write them behind the library's back
 * (test/spmvtest1.c:215, test/test1.c, every driver that prints a solution).  Vectors are ALWAYS library-allocated
 * (src/vector/lis_vector.c:116-156 create, :370-441 duplicate), so the library can give value[] its own pages and let the
 * MMU report the accesses the API cannot see:
 *
 *     pages            who holds the data                 what a host access does
 *     read + write     the host array (HBM copy stale)    nothing: plain memory
 *     read only        host array and HBM copy agree      a WRITE faults: pages -> read + write, HBM copy marked stale
 *     no access        the HBM copy (host array stale)    any access faults: HBM -> host, pages -> read only (a write faults once more)
 *
 * A kernel that writes a vector leaves its pages without access and downloads nothing; a kernel that reads a vector uploads it only
 * when the host array was written since (pages read + write), then makes the pages read-only.  A loop of lis_matvec / lis_vector_*
 * calls therefore runs at the speed of LIS_AMD_RESIDENT, and a program that pokes v->value[i] in between still sees -- and changes --
 * the right numbers, at the cost of one fault and one copy of that vector.  This is the default (LIS_AMD_COHERENT).
 *
 * TWO MAPPINGS OF THE SAME PAGES.  The pages belong to an anonymous memory file (memfd); the program sees them through the mapping whose
 * address is v->value and whose protection follows the table above; the library keeps a second mapping of the same pages (the alias) that
 * is always readable and writable.  Data coming home from HBM is written THROUGH THE ALIAS while the program's mapping still has no
 * access, and only when the last byte has landed does the program's mapping become readable.  A program with several threads (an OpenMP
 * loop over x->value right after lis_solve is ordinary Lis user code) therefore cannot see a half-filled array: the first thread to fault
 * brings the vector home, every other thread that touches the vector meanwhile faults too (the pages are still without access), finds the
 * region marked `filling` and waits on a condition variable until the copy is complete.  (Rounds 1-3 opened the program's own mapping
 * BEFORE the copy: a second thread then read stale bytes without faulting.)
 *
 * The fault handler runs synchronously in the thread that touched the page (SIGSEGV / SEGV_ACCERR), which is ordinary user code, never
 * the library itself: every library routine that reads or writes value[] on the host goes through lisp_protect / lisp_vec_home first.
 * It therefore may take the registry lock and call the HIP runtime.  Faults at addresses that are no region's go to whoever handled
 * SIGSEGV before; the disposition is never reset for an address inside a region.
 *
 * The handler is NOT async-signal-safe in the POSIX sense -- it takes a mutex, may wait on a condition variable and calls the HIP runtime -- and does not need to be:
 * it runs only for a synchronous fault of ordinary user code on a protected buffer, never for an asynchronous signal.  What follows from that is a rule for callers:
 * a protected buffer must not be handed to code that touches it WHILE HOLDING a lock the handler's callees need (a malloc arena, the HIP runtime's own locks): in
 * practice, do not pass v->value to another library's internals without lis_amd_vector_sync_host(v) first.  To keep the handler's own footprint small it copies
 * through a stream, pinned buffers and events of its own (lisd_staged_d2h_fault: the one thing the faulting thread puts on the library's stream is an event record that
 * orders the copy behind the work already queued there -- never a copy --, and a hipGraph capture in progress there is waited out, not invalidated) and with plain memcpy -- no OpenMP region is opened from inside it.
 *
 * What page protection cannot do: a SYSTEM CALL that is handed a protected buffer (write(2) / fwrite of a large v->value, MPI_Send,
 * another device's DMA) does not fault, it fails with EFAULT (a short count from fwrite).  Programs that pass v->value to the kernel
 * call lis_amd_vector_sync_host(v) first (read access) / lis_amd_vector_host_modified(v) after (write access), or run with
 * LIS_AMD_COHERENCE=eager / lis_amd_set_coherence(0): every call then uploads its inputs and downloads its outputs and the pages keep
 * full access.  tests/test_host_cpu.py pins both behaviours.  The pages are shared memory: a forked child sees (and writes) the
 * parent's vectors, not a copy -- a process with a live HIP context must not fork and go on using it anyway.
 * When the memory file or the handler cannot be had, vectors fall back to plain memory and eager coherence -- slower, never stale.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include "lis_internal.h"

typedef struct lisp_region {
	char *base;              /* the program's mapping (v->value / A->index ...), a guard page on either side */
	char *alias;             /* the library's mapping of the same pages: always read + write */
	size_t bytes;            /* whole pages */
	int prot;                /* protection of the program's mapping: LISP_RW / LISP_RO / LISP_NONE */
	int filling;             /* a thread is bringing the data home through the alias: everybody else waits on fill_done */
	LIS_VECTOR owner;        /* a vector's value[] ... */
	/* ... or an array of a matrix the library converted in HBM: the host array is materialised from `dev` on its first touch */
	void *mowner;            /* the matrix */
	void *dev;               /* device buffer holding the array (NULL once the host pages hold it) */
	size_t used;             /* bytes of the array */
	int own_dev;             /* the buffer exists only to back these pages: freed once they are filled (else it is part of the matrix's HBM copy) */
	/* ... or an array handed out by lis_matrix_malloc_<fmt> (lisp_alloc_tracked): plain memory until a matrix adopts it (mowner) and uploads it; from then on
	 * read-only, and the first host write marks the matrix's HBM copy stale (the reference reads the adopted arrays live on every product, lis_matvec_csr.c:97-109) */
	int tracked;
	/* test hook (lis_amd_vector_page_test_source): a host buffer standing in for the HBM copy, copied in two halves `test_delay_ms` apart */
	const double *test_src;
	int test_delay_ms;       /* > 0: that pause between the halves; < 0: the second half waits until -test_delay_ms other threads wait for the copy */
	int waiters;             /* threads that found the current copy in flight */
	struct lisp_region *next;
} lisp_region;

static lisp_region *regions;
static pthread_mutex_t region_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t fill_done = PTHREAD_COND_INITIALIZER;
static struct sigaction previous_action;
static int handler_installed, handler_failed;
static long faults_read, faults_write, faults_waited;
static __thread char *tl_last_fault;
static __thread int tl_repeats;

static size_t page_size(void)
{
	static size_t ps;
	if (!ps) { long v = sysconf(_SC_PAGESIZE); ps = v > 0 ? (size_t)v : 4096; }
	return ps;
}

static int native_prot(int prot) { return prot == LISP_RW ? (PROT_READ | PROT_WRITE) : prot == LISP_RO ? PROT_READ : PROT_NONE; }

/* ---- the copy that brings a region home.  Called with r->filling set by the caller (under region_lock) and the lock RELEASED; the
 * program's mapping keeps whatever protection it has (none, on every path that matters) until the data is complete. */
static LIS_INT region_fill(lisp_region *r, int prot_after, int from_fault)
{
	LIS_INT (*d2h)(void *, const void *, size_t) = from_fault ? lisd_staged_d2h_fault : lisd_staged_d2h;
	LIS_INT err = LIS_SUCCESS;
	if (r->owner) {
		LIS_VECTOR v = r->owner;
		lisd_vec *d = VDEV(v);
		if (r->test_src) {                                    /* tests: a host buffer plays the HBM copy, slowly */
			const size_t half = d->hlen / 2;
			memcpy(r->alias, r->test_src, half * sizeof(double));
			if (r->test_delay_ms > 0) usleep((useconds_t)r->test_delay_ms * 1000);
			else if (r->test_delay_ms < 0) {                  /* hold the copy until that many other threads wait for it (deterministic tests; 10 s at most) */
				for (int ms = 0; ms < 10000; ms++) {
					pthread_mutex_lock(&region_lock);
					const int enough = r->waiters >= -r->test_delay_ms;
					pthread_mutex_unlock(&region_lock);
					if (enough) break;
					usleep(1000);
				}
			}
			memcpy(r->alias + half * sizeof(double), r->test_src + half, (d->hlen - half) * sizeof(double));
		} else if (d->d && d->dev_valid && !d->host_valid) {
			const size_t len = d->hlen < d->cap ? d->hlen : d->cap;
			err = d2h(r->alias, d->d, len * sizeof(double));
		}
		if (!err) d->host_valid = 1;                          /* (a failed copy leaves the pages without access and the HBM copy the truth) */
	} else {
		void *dev = r->dev;
		if (dev && r->used) err = d2h(r->alias, dev, r->used);
		if (!err) {                                           /* (a failed copy keeps its source: the array's only copy) */
			if (dev && r->own_dev) (void)liship_free(dev);
			r->dev = NULL;
		}
	}
	pthread_mutex_lock(&region_lock);
	if (!err && mprotect(r->base, r->bytes, native_prot(prot_after)) == 0) r->prot = prot_after;
	r->filling = 0;
	r->waiters = 0;
	pthread_cond_broadcast(&fill_done);
	pthread_mutex_unlock(&region_lock);
	return err;
}

/* does `matrix` hold an HBM copy that a host write to one of its arrays would leave stale? (lazy coherence only: the other modes never protect) */
static int matrix_copy_live(void *matrix)
{
	return matrix && lisp_lazy() && MDEV((LIS_MATRIX)matrix)->ready && !MDEV((LIS_MATRIX)matrix)->device_only;
}

static void chain_to_previous(int sig, siginfo_t *info, void *context)
{
	if (previous_action.sa_flags & SA_SIGINFO) {
		if (previous_action.sa_sigaction) { previous_action.sa_sigaction(sig, info, context); return; }
	} else if (previous_action.sa_handler == SIG_IGN) {
		return;
	} else if (previous_action.sa_handler != SIG_DFL && previous_action.sa_handler) {
		previous_action.sa_handler(sig);
		return;
	}
	signal(sig, SIG_DFL);                          /* default action: the access is retried and terminates the process */
}

static void on_fault(int sig, siginfo_t *info, void *context)
{
	char *addr = (char *)info->si_addr;
	lisp_region *r = NULL;
	const int saved_errno = errno;
	if (sig == SIGSEGV && addr) {
		pthread_mutex_lock(&region_lock);
		for (r = regions; r; r = r->next) if (addr >= r->base && addr < r->base + r->bytes) break;
		if (r && r->filling) {
			/* another thread is bringing this region home: wait until its copy is complete, then retry the access.  (The region is looked up again after
			 * every wake-up: a thread that destroys the vector waits for the same copy and may have freed the region by the time this one runs.) */
			faults_waited++;
			r->waiters++;
			for (;;) {
				pthread_cond_wait(&fill_done, &region_lock);
				for (r = regions; r; r = r->next) if (addr >= r->base && addr < r->base + r->bytes) break;
				if (!r || !r->filling) break;
			}
			pthread_mutex_unlock(&region_lock);
			tl_last_fault = NULL; tl_repeats = 0;
			errno = saved_errno;
			return;
		}
		if (r && r->prot == LISP_NONE) {
			/* the HBM copy is the truth: bring it home through the alias; the program's mapping opens when the data is there --
			 * read-only for a vector (both sides agree now; a write faults once more) and for an array of a matrix whose HBM copy lives (a write marks
			 * that copy stale), read + write for any other matrix array (plain memory from here on) */
			r->filling = 1;
			faults_read++;
			pthread_mutex_unlock(&region_lock);
			if (lisg.device_ready) (void)liship_set_device(lisg.device);      /* (this thread may never have called the runtime) */
			const LIS_INT err = region_fill(r, (r->owner || matrix_copy_live(r->mowner)) ? LISP_RO : LISP_RW, 1);
			if (err != LIS_SUCCESS) {
				fprintf(stderr, "liblis_amd: could not bring %s back from HBM inside the page-fault handler\n", r->owner ? "a vector" : "a matrix array");
				abort();
			}
			tl_last_fault = NULL; tl_repeats = 0;
			errno = saved_errno;
			return;
		}
		if (r && r->prot == LISP_RO) {
			LIS_VECTOR v = r->owner;
			faults_write++;
			if (mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE) == 0) r->prot = LISP_RW;
			if (v) { VDEV(v)->host_valid = 1; VDEV(v)->dev_valid = 0; }      /* the host array is being written: the HBM copy is stale from here on */
			else if (r->mowner) MDEV((LIS_MATRIX)r->mowner)->host_written = 1;      /* ... the matrix's: rebuilt (arrays, plan) before its next product */
			pthread_mutex_unlock(&region_lock);
			tl_last_fault = NULL; tl_repeats = 0;
			errno = saved_errno;
			return;
		}
		if (r) {
			/* read + write already: another thread resolved this very fault between the access and the lock.  Retry; a fault that keeps
			 * coming back at the same address of writable pages is not ours to resolve (the same thread, 64 times: give it to the previous owner) */
			if (addr == tl_last_fault) tl_repeats++; else { tl_last_fault = addr; tl_repeats = 0; }
			const int give_up = tl_repeats > 64;
			if (!give_up) (void)mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE);
			pthread_mutex_unlock(&region_lock);
			if (!give_up) { errno = saved_errno; return; }
		} else pthread_mutex_unlock(&region_lock);
	}
	/* not ours: the previous disposition decides */
	chain_to_previous(sig, info, context);
	errno = saved_errno;
}

static int install_handler(void)
{
	if (handler_installed) { (void)lisd_fault_stage_prepare(); return 1; }      /* (no-op once made; the device may have come up since the handler was installed) */
	if (handler_failed) return 0;
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_fault;
	sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
	sigemptyset(&sa.sa_mask);
	if (sigaction(SIGSEGV, &sa, &previous_action) != 0) {
		handler_failed = 1;
		lisg.eager_coherence = 1;                  /* no handler, no lazy coherence: every call copies (slower, never stale) */
		fprintf(stderr, "liblis_amd: cannot install the page-fault handler (%s): vectors use eager coherence from here on\n", strerror(errno));
		return 0;
	}
	handler_installed = 1;
	(void)lisd_fault_stage_prepare();              /* the stream and pinned buffers fault-time copies use: made here, in ordinary context */
	{	/* a process-wide signal disposition is not something a library takes silently: one line, once (LIS_AMD_QUIET=1 drops it) */
		const char *q = getenv("LIS_AMD_QUIET");
		if (!(q && q[0] == '1'))
			fprintf(stderr, "liblis_amd: lazy coherence on: a SIGSEGV handler now serves accesses to the library's page-protected vectors and matrix arrays "
			                "(other faults go to the previous handler; LIS_AMD_COHERENCE=eager or LIS_AMD_RESIDENCY=resident run without it)\n");
	}
	return 1;
}

/* A handler the PROGRAM installs later (sigaction(SIGSEGV, ...) of its own) would silently take the faults of protected pages away.  Checked where it is cheap --
 * at the start of every lis_solve and every 1024th change of a protection --: when the disposition is no longer ours, every page is opened for good, the HBM
 * copies are declared stale where the host may have been written, and the process goes on in eager coherence with a line on stderr. */
static void check_handler_still_ours(void)
{
	struct sigaction cur;
	if (!handler_installed || sigaction(SIGSEGV, NULL, &cur) != 0) return;
	if ((cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_fault) return;
	handler_installed = 0; handler_failed = 1;
	lisg.eager_coherence = 1;
	fprintf(stderr, "liblis_amd: the SIGSEGV handler of lazy coherence was replaced by the program: all protected pages are opened and the library copies on every "
	                "call from here on (eager coherence)\n");
	for (int guard = 0; guard < (1 << 20); guard++) {   /* what only HBM holds comes home first (through the alias; a program's read of it could no longer be served) */
		LIS_VECTOR v = NULL; void *m = NULL;
		pthread_mutex_lock(&region_lock);
		for (lisp_region *r = regions; r && !v && !m; r = r->next)
			if (!r->filling && r->prot == LISP_NONE) { if (r->owner) v = r->owner; else if (r->mowner && r->dev) m = r->mowner; }
		pthread_mutex_unlock(&region_lock);
		if (v) { if (lisp_vec_home(v) != LIS_SUCCESS) break; }
		else if (m) { if (lisp_fill_matrix(m) != LIS_SUCCESS) break; }
		else break;
	}
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) {
		if (r->filling || r->prot == LISP_RW) continue;
		if (mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE) == 0) {
			r->prot = LISP_RW;
			if (r->owner) { VDEV(r->owner)->host_valid = 1; VDEV(r->owner)->dev_valid = 0; }      /* (nobody would tell us about a write any more) */
			else if (r->mowner) MDEV((LIS_MATRIX)r->mowner)->host_written = 1;
		}
	}
	pthread_mutex_unlock(&region_lock);
}
void lisp_check_handler(void) { check_handler_still_ours(); }
LIS_INT lis_amd_check_fault_handler(void) { check_handler_still_ours(); return handler_installed ? 1 : 0; }

/* `bytes` (whole pages) of fresh zero pages mapped twice: *base with a guard page on either side and protection `prot`, *alias read + write.
 * 0 on success.  The guard pages keep the kernel from merging the program's mapping into one VMA with a neighbour (measured in round 2: a
 * vector sharing a VMA with arrays the runtime had registered made its first mprotect an MMU-notifier invalidation of that registration --
 * 25 ms per stream synchronisation); a shared file mapping would not merge with anonymous memory anyway, the guards also catch overruns. */
static int map_twice(size_t bytes, int prot, char **base, char **alias)
{
	const size_t ps = page_size();
	int fd = (int)syscall(SYS_memfd_create, "lis_amd_vector", 1u /* MFD_CLOEXEC */);
	if (fd < 0) return -1;
	if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return -1; }
	char *m = (char *)mmap(NULL, bytes + 2 * ps, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (m == (char *)MAP_FAILED) { close(fd); return -1; }
	char *p = (char *)mmap(m + ps, bytes, native_prot(prot), MAP_SHARED | MAP_FIXED, fd, 0);
	char *a = p == (char *)MAP_FAILED ? p : (char *)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);                                     /* the mappings keep the pages alive */
	if (p == (char *)MAP_FAILED || a == (char *)MAP_FAILED) {
		if (a != (char *)MAP_FAILED && p != (char *)MAP_FAILED) munmap(a, bytes);
		munmap(m, bytes + 2 * ps);
		return -1;
	}
#ifdef MADV_HUGEPAGE
	(void)madvise(p, bytes, MADV_HUGEPAGE);        /* (takes effect where shmem huge pages are set to `advise`; harmless elsewhere) */
	(void)madvise(a, bytes, MADV_HUGEPAGE);
#endif
	*base = p; *alias = a;
	return 0;
}

static void unmap_twice(lisp_region *r)
{
	munmap(r->base - page_size(), r->bytes + 2 * page_size());      /* (with its guard pages) */
	munmap(r->alias, r->bytes);
}

/* value[] of `doubles` entries on pages of its own, zero-filled (as the calloc it replaces).  When the memory file cannot be had the vector
 * lives in plain calloc'ed memory (region NULL: copies on every call, like eager coherence).  NULL: out of memory */
LIS_SCALAR *lisp_alloc(LIS_VECTOR v, size_t doubles)
{
	const size_t ps = page_size();
	size_t bytes = (doubles > 0 ? doubles : 1) * sizeof(LIS_SCALAR);
	bytes = (bytes + ps - 1) / ps * ps;
	VDEV(v)->region = NULL;
	lisp_region *r = (lisp_region *)calloc(1, sizeof(*r));
	if (!r) return NULL;
	if (map_twice(bytes, LISP_RW, &r->base, &r->alias) != 0) {
		free(r);
		return (LIS_SCALAR *)calloc(doubles > 0 ? doubles : 1, sizeof(LIS_SCALAR));
	}
	r->bytes = bytes; r->prot = LISP_RW; r->owner = v;
	pthread_mutex_lock(&region_lock);
	r->next = regions; regions = r;
	pthread_mutex_unlock(&region_lock);
	VDEV(v)->region = r;
	return (LIS_SCALAR *)r->base;
}

static void unlink_region(lisp_region *r)
{
	pthread_mutex_lock(&region_lock);
	while (r->filling) pthread_cond_wait(&fill_done, &region_lock);
	for (lisp_region **pp = &regions; *pp; pp = &(*pp)->next) if (*pp == r) { *pp = r->next; break; }
	pthread_mutex_unlock(&region_lock);
}

void lisp_free(LIS_VECTOR v)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	if (!r) return;
	unlink_region(r);
	unmap_twice(r);
	free(r);
	VDEV(v)->region = NULL;
}

/* ---- arrays of a matrix that was converted in HBM (lis_convert.c): the Lis API promises host arrays (A->ptr, A->index, A->value ...), but a
 * program that only multiplies never reads them.  They get address space without access, bound to the device buffer that holds their
 * contents; the first touch -- by the program or by a host-side routine of this library -- brings them home (on_fault, through the alias:
 * no thread sees a half-filled array).  NULL: no memory / no handler / no memory file (the caller converts on the host). */
void *lisp_alloc_lazy(void *matrix, size_t bytes_used, void *dev, int own_dev)
{
	const size_t ps = page_size();
	size_t bytes = (bytes_used > 0 ? bytes_used : 1);
	bytes = (bytes + ps - 1) / ps * ps;
	if (!install_handler()) return NULL;
	lisp_region *r = (lisp_region *)calloc(1, sizeof(*r));
	if (!r) return NULL;
	if (map_twice(bytes, LISP_NONE, &r->base, &r->alias) != 0) { free(r); return NULL; }
	r->bytes = bytes; r->prot = LISP_NONE; r->mowner = matrix; r->dev = dev; r->used = bytes_used; r->own_dev = own_dev;
	pthread_mutex_lock(&region_lock);
	r->next = regions; regions = r;
	pthread_mutex_unlock(&region_lock);
	return r->base;
}

/* ---- arrays handed out by lis_matrix_malloc_<fmt> (src/matrix/lis_matrix_csr.c:170 and its siblings): library memory that a program fills, hands to
 * lis_matrix_set_<fmt> (adopted, not copied: lis_matrix_csr.c:98-103) and may keep writing to between solves -- the reference reads it live on every product.
 * They live on pages like a vector's: read + write until the adopting matrix is uploaded, read-only from then on; a host write faults once, opens the pages and
 * marks the HBM copy (arrays, plan, transposed operator) stale, and the next product rebuilds it.  NULL: no memory file (the caller uses malloc; untracked). */
void *lisp_alloc_tracked(size_t bytes_used)
{
	const size_t ps = page_size();
	size_t bytes = (bytes_used > 0 ? bytes_used : 1);
	bytes = (bytes + ps - 1) / ps * ps;
	lisp_region *r = (lisp_region *)calloc(1, sizeof(*r));
	if (!r) return NULL;
	if (map_twice(bytes, LISP_RW, &r->base, &r->alias) != 0) { free(r); return NULL; }
	r->bytes = bytes; r->prot = LISP_RW; r->used = bytes_used; r->tracked = 1;
	pthread_mutex_lock(&region_lock);
	r->next = regions; regions = r;
	pthread_mutex_unlock(&region_lock);
	return r->base;
}

/* `matrix` adopts `array` (lis_matrix_set_<fmt>, assemble): 1 when the array lives on tracked pages */
int lisp_adopt(void *matrix, void *array)
{
	int hit = 0;
	if (!array) return 0;
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next)
		if ((char *)array == r->base && r->tracked) { r->mowner = matrix; hit = 1; break; }
	pthread_mutex_unlock(&region_lock);
	return hit;
}

/* the HBM copy of `matrix` was just built from its host arrays: those that live on pages of the library's become read-only (lazy coherence).
 * Returns how many arrays are now write-protected. */
int lisp_matrix_protect(void *matrix)
{
	int c = 0;
	if (!lisp_lazy() || !install_handler()) return 0;
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) {
		if (r->mowner != matrix || r->filling) continue;
		if (r->prot == LISP_RW && mprotect(r->base, r->bytes, PROT_READ) == 0) r->prot = LISP_RO;
		c += r->prot == LISP_RO;
	}
	pthread_mutex_unlock(&region_lock);
	return c;
}

/* the HBM copy of `matrix` is gone (or the matrix lets go of its arrays: lis_matrix_unset, destroy without is_destroy): its read-only arrays are plain memory again.
 * forget: the arrays no longer belong to it at all */
void lisp_matrix_release(void *matrix, int forget)
{
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) {
		if (r->mowner != matrix) continue;
		if (r->prot == LISP_RO && !r->filling && mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE) == 0) r->prot = LISP_RW;
		if (forget) r->mowner = NULL;
	}
	pthread_mutex_unlock(&region_lock);
}

/* bytes of the array at p when it lives on pages of the library's (made current on the host first), else 0 */
size_t lisp_array_bytes(void *p)
{
	size_t b = 0;
	if (!p) return 0;
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) if ((char *)p == r->base && !r->owner) { b = r->used ? r->used : 1; break; }
	pthread_mutex_unlock(&region_lock);
	return b;
}

/* how many arrays of `matrix` are write-protected right now (tests) */
int lisp_protected_arrays(void *matrix)
{
	int c = 0;
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) c += (r->mowner == matrix && r->prot == LISP_RO);
	pthread_mutex_unlock(&region_lock);
	return c;
}

static lisp_region *region_at(const void *p)
{
	for (lisp_region *r = regions; r; r = r->next) if ((const char *)p == r->base) return r;
	return NULL;
}

/* free() for an array that may live on such pages: 1 when it did (unmapped; a backing buffer nobody else owns is freed too) */
int lisp_free_array(void *p)
{
	if (!p) return 0;
	pthread_mutex_lock(&region_lock);
	lisp_region *r = region_at(p);
	if (r) {
		while (r->filling) pthread_cond_wait(&fill_done, &region_lock);
		for (lisp_region **pp = &regions; *pp; pp = &(*pp)->next) if (*pp == r) { *pp = r->next; break; }
	}
	pthread_mutex_unlock(&region_lock);
	if (!r) return 0;
	if (r->dev && r->own_dev) (void)liship_free(r->dev);
	unmap_twice(r);
	free(r);
	return 1;
}

/* the HBM copy of `matrix` is about to go while the matrix lives on (or a host routine is about to read its arrays from several
 * threads): every array still held there comes home now */
LIS_INT lisp_fill_matrix(void *matrix)
{
	for (;;) {
		pthread_mutex_lock(&region_lock);
		lisp_region *r = regions;
		while (r && !(r->mowner == matrix && (r->filling || (r->prot == LISP_NONE && r->dev)))) r = r->next;
		if (!r) { pthread_mutex_unlock(&region_lock); return LIS_SUCCESS; }
		if (r->filling) {                              /* a faulting thread of the program is at it: wait for that copy, look again */
			while (r->filling) pthread_cond_wait(&fill_done, &region_lock);
			pthread_mutex_unlock(&region_lock);
			continue;
		}
		r->filling = 1;
		pthread_mutex_unlock(&region_lock);
		LISCHK(region_fill(r, LISP_RW, 0));
	}
}

void lisp_reown(void *from, void *to)
{
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) if (r->mowner == from) r->mowner = to;
	pthread_mutex_unlock(&region_lock);
}

/* how many arrays of `matrix` are still held in HBM only (tests) */
int lisp_lazy_arrays(void *matrix)
{
	int c = 0;
	pthread_mutex_lock(&region_lock);
	for (lisp_region *r = regions; r; r = r->next) c += (r->mowner == matrix && r->prot == LISP_NONE && r->dev != NULL);
	pthread_mutex_unlock(&region_lock);
	return c;
}

/* value[] grows to `doubles` entries (old contents kept, new entries zero), whatever memory it lived in; the caller made it current */
LIS_INT lisp_grow(LIS_VECTOR v, size_t doubles)
{
	lisd_vec *d = VDEV(v);
	LIS_SCALAR *old = v->value;
	lisp_region *oldr = (lisp_region *)d->region;
	const size_t have = d->hlen;
	LIS_SCALAR *nv = lisp_alloc(v, doubles);           /* (sets d->region to the new region, or NULL for plain memory) */
	if (!nv) { d->region = oldr; return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)doubles); }
	const void *src = oldr ? (const void *)oldr->alias : (const void *)old;      /* (the alias: readable whatever the old pages' protection) */
	if (src) memcpy(nv, src, sizeof(LIS_SCALAR) * (have < doubles ? have : doubles));
	if (oldr) {
		lisp_region *keep = (lisp_region *)d->region;
		d->region = oldr;
		lisp_free(v);
		d->region = keep;
	} else if (v->is_destroy) free(old);
	v->value = nv;
	d->hlen = doubles;
	return LIS_SUCCESS;
}

/* protection follows the vector's state; a no-op for vectors without pages of their own and in the other coherence modes.  The caller
 * checks lisp_state() where it matters that the protection really is in place (lisd_vec_in, lisd_vec_done). */
void lisp_protect(LIS_VECTOR v, int prot)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	if (!r) return;
	if (prot != LISP_RW && (!lisp_lazy() || !install_handler())) return;
	{ static unsigned calls; if ((++calls & 1023u) == 0) check_handler_still_ours(); }
	if (prot != LISP_RW && !lisp_lazy()) return;         /* (the check may just have switched to eager coherence) */
	pthread_mutex_lock(&region_lock);
	while (r->filling) pthread_cond_wait(&fill_done, &region_lock);      /* a thread of the program is bringing it home: let the copy finish */
	if (r->prot != prot && mprotect(r->base, r->bytes, native_prot(prot)) == 0) r->prot = prot;
	pthread_mutex_unlock(&region_lock);
}

int lisp_state(LIS_VECTOR v)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	return r ? r->prot : -1;
}

/* value[] current on the host, for a vector on pages of its own under lazy coherence: the HBM copy comes home through the alias (the
 * program's mapping opens, read-only, when the data is there).  Called by the library's own host-side readers / writers
 * (lisd_vec_to_host); a thread of the program that faults on the same vector meanwhile waits for this copy, and vice versa. */
LIS_INT lisp_vec_home(LIS_VECTOR v)
{
	lisd_vec *d = VDEV(v);
	lisp_region *r = (lisp_region *)d->region;
	pthread_mutex_lock(&region_lock);
	while (r->filling) pthread_cond_wait(&fill_done, &region_lock);
	const int stale = r->test_src ? !d->host_valid : (!d->host_valid && d->dev_valid && d->d && v->value);
	if (!stale) {
		d->host_valid = 1;
		if (r->prot == LISP_NONE) {
			const int want = d->dev_valid ? LISP_RO : LISP_RW;
			if (mprotect(r->base, r->bytes, native_prot(want)) == 0) r->prot = want;
		}
		pthread_mutex_unlock(&region_lock);
		return LIS_SUCCESS;
	}
	r->filling = 1;
	pthread_mutex_unlock(&region_lock);
	return region_fill(r, LISP_RO, 0);
}

/* lazy coherence applies to the COHERENT residency unless switched off */
int lisp_lazy(void) { return lisg.residency == LIS_AMD_COHERENT && !lisg.eager_coherence; }

LIS_INT lis_amd_set_coherence(LIS_INT lazy)
{
	lisg.eager_coherence = (lazy && !handler_failed) ? 0 : 1;
	return LIS_SUCCESS;
}

LIS_INT lis_amd_matrix_lazy_arrays(LIS_MATRIX A) { return lisp_lazy_arrays(A); }
LIS_INT lis_amd_matrix_protected_arrays(LIS_MATRIX A) { return lisp_protected_arrays(A); }
LIS_INT lis_amd_set_device_convert(LIS_INT on) { lisg.no_device_convert = on ? 0 : 1; return LIS_SUCCESS; }
LIS_INT lis_amd_vector_page_state(LIS_VECTOR v) { return lisp_state(v); }
LIS_INT lis_amd_vector_page_protect(LIS_VECTOR v, LIS_INT state) { lisp_protect(v, (int)state); return lisp_state(v) == (int)state ? LIS_SUCCESS : LIS_ERR_ILL_ARG; }

/* tests of the fault handler without a GPU: `src` (n + pad doubles that must outlive the vector's next fault) plays the HBM copy of v --
 * the pages lose all access, and the first touch copies src home in two halves delay_ms apart, through the alias.  NULL removes the hook. */
LIS_INT lis_amd_vector_page_test_source(LIS_VECTOR v, const LIS_SCALAR *src, LIS_INT delay_ms)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	if (!r) return LISI_ERR(LIS_ERR_ILL_ARG, "vector v has no pages of its own\n");
	pthread_mutex_lock(&region_lock);
	while (r->filling) pthread_cond_wait(&fill_done, &region_lock);
	r->test_src = src; r->test_delay_ms = (int)delay_ms;
	pthread_mutex_unlock(&region_lock);
	if (!src) return LIS_SUCCESS;
	VDEV(v)->host_valid = 0; VDEV(v)->dev_valid = 1;
	lisp_protect(v, LISP_NONE);
	return lisp_state(v) == LISP_NONE ? LIS_SUCCESS : LIS_ERR_NOT_IMPLEMENTED;
}

LIS_INT lis_amd_page_faults(LIS_INT *reads, LIS_INT *writes)
{
	if (reads) *reads = (LIS_INT)faults_read;
	if (writes) *writes = (LIS_INT)faults_write;
	return LIS_SUCCESS;
}
LIS_INT lis_amd_page_fault_waits(void) { return (LIS_INT)faults_waited; }
