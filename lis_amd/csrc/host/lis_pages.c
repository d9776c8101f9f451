/*
 * lis_pages.c -- coherent semantics at resident speed: the host pages of a vector follow its HBM copy through page protection.
 *
 * Lis hands out raw host arrays (v->value[], lis.h:513-537) and programs read and write them behind the library's back
 * (test/spmvtest1.c:215, test/test1.c, every driver that prints a solution).  Vectors are ALWAYS library-allocated
 * (src/vector/lis_vector.c:116-156 create, :370-441 duplicate), so the library can give value[] its own pages (mmap) and let the
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
 * the right numbers, at the cost of one fault and one copy of that vector.  This is the default (LIS_AMD_COHERENT);
 * LIS_AMD_COHERENCE=eager (or lis_amd_set_coherence(0)) restores the copy-on-every-call behaviour, for programs that hand v->value to
 * something the MMU cannot interrupt: a system call (write(2) of a protected buffer fails with EFAULT instead of faulting) or another
 * device's DMA.  Such code can also bracket the access with lis_amd_vector_sync_host() / lis_amd_vector_host_modified().
 *
 * The fault handler runs synchronously in the thread that touched the page (SIGSEGV / SEGV_ACCERR), which is ordinary user code, never
 * the library itself: every library routine that reads or writes value[] on the host unprotects first (lisd_vec_to_host,
 * lisd_vec_host_write).  It therefore may take the registry lock and call the HIP runtime.  Faults at addresses that are not a vector's
 * go to whoever handled SIGSEGV before.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
#include <sys/mman.h>
#include "lis_internal.h"

typedef struct lisp_region {
	char *base;
	size_t bytes;            /* whole pages */
	int prot;                /* LISP_RW / LISP_RO / LISP_NONE */
	LIS_VECTOR owner;        /* a vector's value[] ... */
	/* ... or an array of a matrix the library converted in HBM: the host array is materialised from `dev` on its first touch */
	void *mowner;            /* the matrix */
	void *dev;               /* device buffer holding the array (NULL once the host pages hold it) */
	size_t used;             /* bytes of the array */
	int own_dev;             /* the buffer exists only to back these pages: freed once they are filled (else it is part of the matrix's HBM copy) */
	struct lisp_region *next;
} lisp_region;

static lisp_region *regions;
static pthread_mutex_t region_lock = PTHREAD_MUTEX_INITIALIZER;
static struct sigaction previous_action;
static int handler_installed;
static long faults_read, faults_write;
static pthread_mutex_t sync_lock = PTHREAD_MUTEX_INITIALIZER;      /* held while a vector travels HBM -> host inside the handler */
static char *last_rw_fault;

static size_t page_size(void)
{
	static size_t ps;
	if (!ps) { long v = sysconf(_SC_PAGESIZE); ps = v > 0 ? (size_t)v : 4096; }
	return ps;
}

static int native_prot(int prot) { return prot == LISP_RW ? (PROT_READ | PROT_WRITE) : prot == LISP_RO ? PROT_READ : PROT_NONE; }

static void on_fault(int sig, siginfo_t *info, void *context)
{
	char *addr = (char *)info->si_addr;
	lisp_region *r = NULL;
	if (sig == SIGSEGV && addr) {
		pthread_mutex_lock(&region_lock);
		for (r = regions; r; r = r->next) if (addr >= r->base && addr < r->base + r->bytes) break;
		if (r && r->prot == LISP_NONE && !r->owner) {
			/* an array of a matrix converted in HBM, touched for the first time: it becomes plain host memory */
			faults_read++;
			pthread_mutex_lock(&sync_lock);
			mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE);
			r->prot = LISP_RW;
			void *dev = r->dev;
			const size_t used = r->used;
			const int own = r->own_dev;
			r->dev = NULL;
			pthread_mutex_unlock(&region_lock);
			LIS_INT err = LIS_SUCCESS;
			if (dev && used) err = lisd_staged_d2h(r->base, dev, used);
			if (dev && own) (void)liship_free(dev);
			pthread_mutex_unlock(&sync_lock);
			if (err != LIS_SUCCESS) {
				fprintf(stderr, "liblis_amd: could not bring a matrix array back from HBM inside the page-fault handler\n");
				abort();
			}
			return;
		}
		if (r && r->prot == LISP_NONE) {
			/* the HBM copy is the truth: bring it home, leave the pages read-only (both sides agree now) */
			LIS_VECTOR v = r->owner;
			faults_read++;
			pthread_mutex_lock(&sync_lock);        /* (a second thread that faults on these pages meanwhile waits below until the data is there) */
			mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE);
			r->prot = LISP_RW;
			pthread_mutex_unlock(&region_lock);
			const LIS_INT err = lisd_vec_to_host(v);
			pthread_mutex_unlock(&sync_lock);
			if (err != LIS_SUCCESS) {
				fprintf(stderr, "liblis_amd: could not bring a vector back from HBM inside the page-fault handler\n");
				abort();
			}
			return;                                /* the access is retried; a write now faults on the read-only pages */
		}
		if (r && r->prot == LISP_RO) {
			LIS_VECTOR v = r->owner;
			faults_write++;
			mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE);
			r->prot = LISP_RW;
			VDEV(v)->host_valid = 1;
			VDEV(v)->dev_valid = 0;                /* the host array is being written: the HBM copy is stale from here on */
			pthread_mutex_unlock(&region_lock);
			return;
		}
		if (r && addr != last_rw_fault) {          /* another thread resolved (or is resolving) this very fault: wait for its copy, retry */
			last_rw_fault = addr;
			pthread_mutex_unlock(&region_lock);
			pthread_mutex_lock(&sync_lock);
			pthread_mutex_unlock(&sync_lock);
			return;
		}
		pthread_mutex_unlock(&region_lock);
	}
	/* not ours (or a fault that repeats inside a writable vector): the previous disposition decides */
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

static int install_handler(void)
{
	if (handler_installed) return 1;
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_fault;
	sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
	sigemptyset(&sa.sa_mask);
	if (sigaction(SIGSEGV, &sa, &previous_action) != 0) return 0;
	handler_installed = 1;
	return 1;
}

/* value[] of `doubles` entries on pages of its own, zero-filled (as the calloc it replaces); NULL: out of memory */
LIS_SCALAR *lisp_alloc(LIS_VECTOR v, size_t doubles)
{
	const size_t ps = page_size();
	size_t bytes = (doubles > 0 ? doubles : 1) * sizeof(LIS_SCALAR);
	bytes = (bytes + ps - 1) / ps * ps;
	/* an inaccessible guard page on either side keeps the kernel from merging these pages into one mapping (VMA) with a neighbour.  Measured
	 * without them: the runtime registers the malloc'ed arrays of a matrix with the GPU driver while it uploads them (a pageable copy pins
	 * its source); a vector mapped next to them shares their VMA, and the first mprotect of the vector then invalidates that registration --
	 * the driver evicts the process's queues and the next stream synchronisation takes 25 ms (test/spmvtest3.c creates its vectors before it
	 * converts its matrix: 35 instead of 7 ms for 100 products at 200^3) */
	char *m = (char *)mmap(NULL, bytes + 2 * ps, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (m == (char *)MAP_FAILED) return NULL;
	void *p = m + ps;
	if (mprotect(p, bytes, PROT_READ | PROT_WRITE) != 0) { munmap(m, bytes + 2 * ps); return NULL; }
	lisp_region *r = (lisp_region *)malloc(sizeof(*r));
	if (!r) { munmap(m, bytes + 2 * ps); return NULL; }
	memset(r, 0, sizeof(*r));
	r->base = (char *)p; r->bytes = bytes; r->prot = LISP_RW; r->owner = v;
	pthread_mutex_lock(&region_lock);
	r->next = regions; regions = r;
	pthread_mutex_unlock(&region_lock);
	VDEV(v)->region = r;
	return (LIS_SCALAR *)p;
}

void lisp_free(LIS_VECTOR v)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	if (!r) return;
	pthread_mutex_lock(&region_lock);
	for (lisp_region **pp = &regions; *pp; pp = &(*pp)->next) if (*pp == r) { *pp = r->next; break; }
	pthread_mutex_unlock(&region_lock);
	munmap(r->base - page_size(), r->bytes + 2 * page_size());      /* (with its guard pages) */
	free(r);
	VDEV(v)->region = NULL;
}

/* ---- arrays of a matrix that was converted in HBM (lis_convert.c): the Lis API promises host arrays (A->ptr, A->index, A->value ...), but a
 * program that only multiplies never reads them.  They get address space without access, bound to the device buffer that holds their
 * contents; the first touch -- by the program or by a host-side routine of this library -- brings them home (on_fault).  NULL: no memory. */
void *lisp_alloc_lazy(void *matrix, size_t bytes_used, void *dev, int own_dev)
{
	const size_t ps = page_size();
	size_t bytes = (bytes_used > 0 ? bytes_used : 1);
	bytes = (bytes + ps - 1) / ps * ps;
	if (!install_handler()) return NULL;
	char *m = (char *)mmap(NULL, bytes + 2 * ps, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (m == (char *)MAP_FAILED) return NULL;
	lisp_region *r = (lisp_region *)calloc(1, sizeof(*r));
	if (!r) { munmap(m, bytes + 2 * ps); return NULL; }
	r->base = m + ps; r->bytes = bytes; r->prot = LISP_NONE; r->mowner = matrix; r->dev = dev; r->used = bytes_used; r->own_dev = own_dev;
	pthread_mutex_lock(&region_lock);
	r->next = regions; regions = r;
	pthread_mutex_unlock(&region_lock);
	return r->base;
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
	if (r) for (lisp_region **pp = &regions; *pp; pp = &(*pp)->next) if (*pp == r) { *pp = r->next; break; }
	pthread_mutex_unlock(&region_lock);
	if (!r) return 0;
	if (r->dev && r->own_dev) (void)liship_free(r->dev);
	munmap(r->base - page_size(), r->bytes + 2 * page_size());
	free(r);
	return 1;
}

/* the HBM copy of `matrix` is about to go while the matrix lives on: every array still held there comes home now */
LIS_INT lisp_fill_matrix(void *matrix)
{
	for (;;) {
		pthread_mutex_lock(&region_lock);
		lisp_region *r = regions;
		while (r && !(r->mowner == matrix && r->prot == LISP_NONE && r->dev)) r = r->next;
		if (!r) { pthread_mutex_unlock(&region_lock); return LIS_SUCCESS; }
		mprotect(r->base, r->bytes, PROT_READ | PROT_WRITE);
		r->prot = LISP_RW;
		void *dev = r->dev;
		const size_t used = r->used;
		const int own = r->own_dev;
		r->dev = NULL;
		pthread_mutex_unlock(&region_lock);
		LIS_INT err = used ? lisd_staged_d2h(r->base, dev, used) : LIS_SUCCESS;
		if (own) (void)liship_free(dev);
		if (err) return err;
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
	d->region = NULL;
	LIS_SCALAR *nv = lisp_alloc(v, doubles);
	if (!nv) { d->region = oldr; return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", (LIS_INT)doubles); }
	if (oldr && oldr->prot == LISP_NONE) mprotect(oldr->base, oldr->bytes, PROT_READ);
	if (old) memcpy(nv, old, sizeof(LIS_SCALAR) * (have < doubles ? have : doubles));
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

/* protection follows the vector's state; a no-op for vectors without pages of their own and in the other coherence modes */
void lisp_protect(LIS_VECTOR v, int prot)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	if (!r || r->prot == prot) return;
	if (prot != LISP_RW && (!lisp_lazy() || !install_handler())) return;
	pthread_mutex_lock(&region_lock);
	if (mprotect(r->base, r->bytes, native_prot(prot)) == 0) r->prot = prot;
	pthread_mutex_unlock(&region_lock);
}

int lisp_state(LIS_VECTOR v)
{
	lisp_region *r = (lisp_region *)VDEV(v)->region;
	return r ? r->prot : -1;
}

/* lazy coherence applies to the COHERENT residency unless switched off */
int lisp_lazy(void) { return lisg.residency == LIS_AMD_COHERENT && !lisg.eager_coherence; }

LIS_INT lis_amd_set_coherence(LIS_INT lazy)
{
	lisg.eager_coherence = lazy ? 0 : 1;
	return LIS_SUCCESS;
}

LIS_INT lis_amd_matrix_lazy_arrays(LIS_MATRIX A) { return lisp_lazy_arrays(A); }
LIS_INT lis_amd_set_device_convert(LIS_INT on) { lisg.no_device_convert = on ? 0 : 1; return LIS_SUCCESS; }
LIS_INT lis_amd_vector_page_state(LIS_VECTOR v) { return lisp_state(v); }
LIS_INT lis_amd_vector_page_protect(LIS_VECTOR v, LIS_INT state) { lisp_protect(v, (int)state); return lisp_state(v) == (int)state ? LIS_SUCCESS : LIS_ERR_ILL_ARG; }

LIS_INT lis_amd_page_faults(LIS_INT *reads, LIS_INT *writes)
{
	if (reads) *reads = (LIS_INT)faults_read;
	if (writes) *writes = (LIS_INT)faults_write;
	return LIS_SUCCESS;
}
