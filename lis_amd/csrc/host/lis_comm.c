/*
 * lis_comm.c -- the distributed layer: row-block partition, ghost renumbering, halo tables, halo
 * exchange and cross-rank reduction folds.  One process per GPU; RCCL over xGMI on the data path.
 *
 * Specification = the reference's MPI layer (src/matrix/lis_matrix_mpi.c):
 *   ranges        lis_ranges_create, src/system/lis_init.c:405-473 (LIS_GET_ISIE split, or given local sizes)
 *   g2l           lis_matrix_g2l_csr :222-320     owned col g -> g-is; ghosts -> n.. in ascending global order
 *   commtable     lis_commtable_create :594-828   neighbours ascending, import slots n..np-1 contiguous per owner
 *   halo          lis_send_recv :834-955          pack ws[i]=x[export_index[i]], exchange, land in x[n..np)
 *   reductions    MPI_Allreduce(1 scalar) lis_vector_ops.c:119,263
 * What is different on MI355X: the exchange is ncclSend/ncclRecv grouped per neighbour, queued on the
 * library's HIP stream, receiving STRAIGHT into x[n + import_ptr[i]) (ghost slots are contiguous per
 * owner, so no unpack pass), and a reduction is an all-gather of the per-rank partials folded in rank
 * order on every rank -- deterministic, unlike a ring all-reduce.  RCCL is loaded at run time with
 * dlopen so single-GPU users never need it.
 *
 * A second backend routes the same calls through host-memory callbacks (lis_amd_comm_init_callbacks):
 * it exists so the partition / table / exchange logic is testable on CPU with torch.distributed+gloo.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include "lis_internal.h"

/* ------------------------------------------------------------------ RCCL, bound at run time */
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
enum { NCCL_DOUBLE = 8, NCCL_INT8 = 0 };
static struct {
	void *dl;
	int (*GetUniqueId)(nccl_uid *);
	int (*CommInitRank)(nccl_comm *, int, nccl_uid, int);
	int (*CommDestroy)(nccl_comm);
	int (*AllGather)(const void *, void *, size_t, int, nccl_comm, void *);
	int (*Send)(const void *, size_t, int, int, nccl_comm, void *);
	int (*Recv)(void *, size_t, int, int, nccl_comm, void *);
	int (*GroupStart)(void);
	int (*GroupEnd)(void);
	const char *(*GetErrorString)(int);
} rccl;

static LIS_INT rccl_load(void)
{
	if (rccl.dl) return LIS_SUCCESS;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", NULL};
	for (int i = 0; names[i] && !rccl.dl; i++) rccl.dl = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
	if (!rccl.dl) { fprintf(stderr, "liblis_amd: cannot load librccl.so.1: %s\n", dlerror()); return LIS_ERR_NOT_IMPLEMENTED; }
#define BIND(field, sym) do { *(void **)(&rccl.field) = dlsym(rccl.dl, sym); \
	if (!rccl.field) { fprintf(stderr, "liblis_amd: librccl lacks %s\n", sym); return LIS_ERR_NOT_IMPLEMENTED; } } while (0)
	BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
	BIND(AllGather, "ncclAllGather"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv");
	BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
	return LIS_SUCCESS;
}

static LIS_INT nccl_fail(const char *what, int rc)
{
	fprintf(stderr, "liblis_amd: %s failed: %s\n", what, rccl.GetErrorString ? rccl.GetErrorString(rc) : "?");
	return LIS_AMD_ERR_DEVICE;
}
#define NCCLCHK(call) do { int rc__ = (call); if (rc__ != 0) return nccl_fail(#call, rc__); } while (0)

/* ncclCommInitRank blocks inside RCCL until every rank has joined: a rank that never comes (a crashed peer, a wrong world size) would hang the others for ever.
 * It runs on a helper thread; the caller waits for it with the communication time limit and aborts the process, loudly, when it expires. */
#include <pthread.h>
#include <time.h>
typedef struct { nccl_comm *out; int nprocs, rank, device, rc; nccl_uid id; } init_job;
static void *init_thread(void *p)
{
	init_job *j = (init_job *)p;
	(void)liship_set_device(j->device);
	j->rc = rccl.CommInitRank(j->out, j->nprocs, j->id, j->rank);
	return NULL;
}
static int comm_init_guarded(nccl_comm *out, int nprocs, const nccl_uid *id, int rank, int device)
{
	const char *e = getenv("LIS_AMD_COMM_TIMEOUT");
	const double limit = e ? atof(e) : 300.0;
	init_job j = {out, nprocs, rank, device, -1, *id};
	if (limit <= 0.0) return rccl.CommInitRank(out, nprocs, *id, rank);
	pthread_t th;
	if (pthread_create(&th, NULL, init_thread, &j) != 0) return rccl.CommInitRank(out, nprocs, *id, rank);
	struct timespec until;
	clock_gettime(CLOCK_REALTIME, &until);
	until.tv_sec += (time_t)limit + 1;
	if (pthread_timedjoin_np(th, NULL, &until) != 0) {
		fprintf(stderr, "liblis_amd: rank %d of %d: ncclCommInitRank did not return within %.0f s (LIS_AMD_COMM_TIMEOUT): a peer never joined -- aborting\n", rank, nprocs, limit);
		fflush(stderr);
		abort();
	}
	return j.rc;
}

LIS_INT lis_amd_comm_get_unique_id(void *id128)
{
	LISCHK(rccl_load());
	nccl_uid id;
	NCCLCHK(rccl.GetUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return LIS_SUCCESS;
}

LIS_INT lis_amd_comm_init_rccl(const void *id128, LIS_INT rank, LIS_INT nprocs, LIS_INT device)
{
	if (lisg.comm_kind) return LISI_ERR(LIS_ERR_ILL_ARG, "communicator already initialised\n");
	if (rank < 0 || rank >= nprocs) return LISI_ERR(LIS_ERR_ILL_ARG, "rank %D out of [0,%D)\n", rank, nprocs);
	if (lisg.device_ready && lisg.device != device) return LISI_ERR(LIS_ERR_ILL_ARG, "device already chosen (%D)\n", lisg.device);
	LISCHK(rccl_load());
	lisg.device = device;
	lisg.comm_kind = 1;
	HIPCHK(liship_set_device(device));
	LISCHK(lisd_init());
	nccl_uid id;
	memcpy(&id, id128, sizeof(id));
	nccl_comm c = NULL;
	NCCLCHK(comm_init_guarded(&c, nprocs, &id, rank, device));
	lisg.nccl_comm = c;
	lisg.rank = rank; lisg.nprocs = nprocs;
	HIPCHK(lisd_malloc((void **)&lisg.gather_out, sizeof(double) * 4 * (size_t)nprocs));
	/* from here on a stream that waits for a peer that never arrives is a loud failure, not a hang (liship_set_sync_timeout, lisi_hip_error) */
	{
		const char *e = getenv("LIS_AMD_COMM_TIMEOUT");
		const double limit = e ? atof(e) : 300.0;
		HIPCHK(liship_set_sync_timeout(limit));
	}
	/* A communicator of its own for the halo exchange.  The overlapped product queues ncclSend/ncclRecv on the second stream while the folds' all-gathers run
	 * on the library's stream (lisc_halo_begin / lisc_fold): two streams on ONE communicator are legal only as long as every operation is ordered against
	 * every other, which the events do -- but it is the kind of arrangement that deadlocks on first contact with new hardware.  With two communicators the two
	 * streams share nothing.  Its unique id is made on rank 0 and travels over the first communicator; if it cannot be formed on EVERY rank (an all-gathered
	 * flag) the job goes on with the one communicator and says so. */
	if (nprocs > 1 && !(getenv("LIS_AMD_ONE_COMMUNICATOR") && getenv("LIS_AMD_ONE_COMMUNICATOR")[0] == '1')) {
		nccl_uid hid, *all = (nccl_uid *)malloc(sizeof(nccl_uid) * (size_t)nprocs);
		int ok = all != NULL;
		memset(&hid, 0, sizeof(hid));
		if (ok && rank == 0) ok = rccl.GetUniqueId(&hid) == 0;
		if (all && lisc_allgather_host(&hid, all, sizeof(hid)) != LIS_SUCCESS) ok = 0;      /* (every rank takes part whatever its own state) */
		nccl_comm h = NULL;
		if (ok) {
			/* rank 0's GetUniqueId may have failed: its id then arrives as zeros on EVERY rank, and every rank skips the second communicator together
			 * (a CommInitRank on a zero id would block until the watchdog) */
			static const nccl_uid zero_id;
			hid = all[0];
			ok = memcmp(&hid, &zero_id, sizeof(hid)) != 0;
		}
		if (ok) ok = comm_init_guarded(&h, nprocs, &hid, rank, device) == 0;
		free(all);
		int mine = ok, *flags = (int *)calloc((size_t)nprocs, sizeof(int));
		if (flags && lisc_allgather_host(&mine, flags, sizeof(int)) == LIS_SUCCESS) { for (LIS_INT r = 0; r < nprocs; r++) ok = ok && flags[r]; } else ok = 0;
		free(flags);
		if (ok) lisg.nccl_halo = h;
		else {
			if (h) (void)rccl.CommDestroy(h);
			if (rank == 0) fprintf(stderr, "liblis_amd: no second RCCL communicator for the halo exchange: sharing the first one (ordered by events)\n");
		}
	}
	return LIS_SUCCESS;
}

LIS_INT lis_amd_comm_init_callbacks(const lis_amd_comm_callbacks *cb, LIS_INT rank, LIS_INT nprocs)
{
	if (lisg.comm_kind) return LISI_ERR(LIS_ERR_ILL_ARG, "communicator already initialised\n");
	if (!cb || !cb->allgather || !cb->neighbor_exchange) return LISI_ERR(LIS_ERR_ILL_ARG, "callbacks missing\n");
	lisg.cb = *cb;
	lisg.comm_kind = 2;
	lisg.rank = rank; lisg.nprocs = nprocs;
	return LIS_SUCCESS;
}

LIS_INT lis_amd_comm_finalize(void)
{
	if (lisg.comm_kind == 1 && lisg.nccl_comm) {
		(void)liship_device_synchronize();
		if (lisg.nccl_halo) (void)rccl.CommDestroy(lisg.nccl_halo);
		(void)rccl.CommDestroy(lisg.nccl_comm);
		(void)liship_set_sync_timeout(0.0);
	}
	if (lisg.comm_stream) {
		(void)liship_event_destroy(lisg.ev_packed); (void)liship_event_destroy(lisg.ev_landed);
		(void)liship_stream_destroy(lisg.comm_stream);
		lisg.comm_stream = lisg.ev_packed = lisg.ev_landed = NULL;
	}
	if (lisg.gather_out) { (void)liship_free(lisg.gather_out); lisg.gather_out = NULL; }
	lisg.nccl_comm = lisg.nccl_halo = NULL; lisg.comm_kind = 0; lisg.rank = 0; lisg.nprocs = 1;
	return LIS_SUCCESS;
}

LIS_INT lis_amd_comm_rank(void) { return lisg.rank; }
LIS_INT lis_amd_comm_kind(void) { return lisg.comm_kind; }
LIS_INT lis_amd_comm_halo_communicator(void) { return lisg.nccl_halo != NULL; }      /* 1: the overlapped halo exchange has an RCCL communicator of its own */
LIS_INT lis_amd_set_overlap(LIS_INT on) { lisg.no_overlap = on ? 0 : 1; return LIS_SUCCESS; }
/* the halo exchange of one product on its own (pack, send/recv, ghosts landed), for timing it apart from the rows */
LIS_INT lis_amd_halo_exchange(LIS_MATRIX A, LIS_VECTOR x)
{
	double *dx;
	LISCHK(lisd_mat_ready(A));
	LISCHK(lisd_vec_in(x, &dx));
	if (lisg.nprocs > 1 && A->commtable) LISCHK(lisc_halo_device(A, dx));
	return LIS_SUCCESS;
}
LIS_INT lis_amd_comm_size(void) { return lisg.nprocs ? lisg.nprocs : 1; }

/* every rank contributes `bytes`; recv holds nprocs*bytes in rank order (host memory) */
LIS_INT lisc_allgather_host(const void *send, void *recv, size_t bytes)
{
	if (lisg.nprocs <= 1) { memcpy(recv, send, bytes); return LIS_SUCCESS; }
	if (lisg.comm_kind == 2) return lisg.cb.allgather(lisg.cb.ctx, send, recv, bytes) ? LIS_ERR_NOT_IMPLEMENTED : LIS_SUCCESS;
	if (lisg.comm_kind == 1) {                  /* setup-time only: stage through HBM */
		void *ds = NULL, *dr = NULL;
		HIPCHK(lisd_malloc(&ds, bytes)); HIPCHK(lisd_malloc(&dr, bytes * (size_t)lisg.nprocs));
		HIPCHK(liship_memcpy_h2d(ds, send, bytes, lisg.stream));
		NCCLCHK(rccl.AllGather(ds, dr, bytes, NCCL_INT8, lisg.nccl_comm, lisg.stream));
		HIPCHK(liship_memcpy_d2h(recv, dr, bytes * (size_t)lisg.nprocs, lisg.stream));
		HIPCHK(liship_stream_synchronize(lisg.stream));
		(void)liship_free(ds); (void)liship_free(dr);
		return LIS_SUCCESS;
	}
	return LISI_ERR(LIS_ERR_ILL_ARG, "nprocs > 1 without a communicator\n");
}

/* the device half of a fold: gather every rank's partial sums into HBM, where the next scalar step adds them in
 * rank order (liship_krylov_step) -- no host synchronisation */
LIS_INT lisc_gather_device(const double *src, int count)
{
	if (lisg.comm_kind != 1 || count > 4) return LISI_ERR(LIS_ERR_ILL_ARG, "device gather needs the RCCL communicator\n");
	NCCLCHK(rccl.AllGather(src, lisg.gather_out, (size_t)count, NCCL_DOUBLE, lisg.nccl_comm, lisg.stream));
	return LIS_SUCCESS;
}

/* sum over ranks in rank order; `inout` holds this rank's partials on entry */
LIS_INT lisc_fold(int count, double *inout)
{
	if (lisg.nprocs <= 1) return LIS_SUCCESS;
	double stack_all[4 * 64], *all = stack_all;
	if (count > 4 || lisg.nprocs > 64) return LISI_ERR(LIS_ERR_ILL_ARG, "fold of %D values over %D ranks\n", count, lisg.nprocs);
	if (lisg.comm_kind == 1 && lisg.host_out) all = lisg.host_out;       /* page-locked landing zone */
	if (lisg.comm_kind == 1) {
		/* reduce_out already holds the partials in HBM: gather them device-side, one small D2H */
		NCCLCHK(rccl.AllGather(lisg.reduce_out, lisg.gather_out, (size_t)count, NCCL_DOUBLE, lisg.nccl_comm, lisg.stream));
		HIPCHK(liship_memcpy_d2h(all, lisg.gather_out, sizeof(double) * (size_t)count * (size_t)lisg.nprocs, lisg.stream));
		HIPCHK(liship_stream_synchronize(lisg.stream));
	} else {
		LISCHK(lisc_allgather_host(inout, all, sizeof(double) * (size_t)count));
	}
	for (int k = 0; k < count; k++) {
		double s = 0.0;
		for (int r = 0; r < lisg.nprocs; r++) s += all[r * count + k];
		inout[k] = s;
	}
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ partition */
LIS_INT lisc_ranges_create(LIS_Comm comm, LIS_INT *local_n, LIS_INT *global_n, LIS_INT **ranges,
                           LIS_INT *is, LIS_INT *ie, LIS_INT *nprocs, LIS_INT *my_rank)
{
	(void)comm;
	const LIS_INT P = lisg.nprocs ? lisg.nprocs : 1, me = lisg.rank;
	*nprocs = P; *my_rank = me;
	if (P == 1) {
		*ranges = NULL;
		if (*local_n == 0) { *local_n = *global_n; } else { *global_n = *local_n; }
		*is = 0; *ie = *local_n;
		return LIS_SUCCESS;
	}
	LIS_INT *r = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(P + 1));
	if (!r) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", P + 1);
	LIS_INT *locals = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)P);
	LIS_INT err = lisc_allgather_host(local_n, locals, sizeof(LIS_INT));
	if (err) { free(r); free(locals); return err; }
	LIS_INT total = 0;
	for (LIS_INT p = 0; p < P; p++) total += locals[p];
	r[0] = 0;
	if (total == 0) {                           /* sizes by the static split of the global size */
		for (LIS_INT p = 0; p < P; p++) { LIS_INT s, e; LIS_GET_ISIE(p, P, *global_n, s, e); (void)s; r[p + 1] = e; }
	} else {
		for (LIS_INT p = 0; p < P; p++) r[p + 1] = r[p] + locals[p];
		*global_n = r[P];
	}
	free(locals);
	*is = r[me]; *ie = r[me + 1];
	*local_n = *ie - *is;
	*ranges = r;
	return LIS_SUCCESS;
}

/* global -> local columns of a CSR matrix; ghost columns numbered n, n+1, ... in ascending global order */
LIS_INT lisc_matrix_g2l(LIS_MATRIX A)
{
	const LIS_INT n = A->n, is = A->is, ie = A->ie, nnz = A->ptr ? A->ptr[n] : 0;
	LIS_INT nghost_refs = 0;
	for (LIS_INT k = 0; k < nnz; k++) if (A->index[k] < is || A->index[k] >= ie) nghost_refs++;
	LIS_INT *g = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nghost_refs > 0 ? nghost_refs : 1));
	if (!g) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", nghost_refs);
	LIS_INT w = 0;
	for (LIS_INT k = 0; k < nnz; k++) if (A->index[k] < is || A->index[k] >= ie) g[w++] = A->index[k];
	/* sorted unique list of ghost globals (heap sort of an int array through the row sorter's twin) */
	if (w > 1) {
		LIS_SCALAR *dummy = (LIS_SCALAR *)calloc((size_t)w, sizeof(LIS_SCALAR));
		lisi_sort_row(0, w, g, dummy);
		free(dummy);
	}
	LIS_INT ns = 0;
	for (LIS_INT k = 0; k < w; k++) if (k == 0 || g[k] != g[k - 1]) g[ns++] = g[k];
	for (LIS_INT k = 0; k < nnz; k++) {
		const LIS_INT c = A->index[k];
		if (c >= is && c < ie) A->index[k] = c - is;
		else {                                   /* rank of c among the ghosts */
			LIS_INT lo = 0, hi = ns;
			while (lo < hi) { LIS_INT mid = (lo + hi) / 2; if (g[mid] < c) lo = mid + 1; else hi = mid; }
			A->index[k] = n + lo;
		}
	}
	A->np = n + ns;
	A->l2g_map = (LIS_INT *)realloc(g, sizeof(LIS_INT) * (size_t)(ns > 0 ? ns : 1));
	return LIS_SUCCESS;
}

void lisc_commtable_destroy(LIS_COMMTABLE t)
{
	if (!t) return;
	free(t->neibpe); free(t->import_ptr); free(t->import_index); free(t->export_ptr); free(t->export_index);
	free(t->ws); free(t->wr);
	free(t);
}

/* neighbour lists from the ghost maps of all ranks: rank r imports l2g_r[]; what r imports from me is what
 * I export to r, in r's (ascending global) order */
LIS_INT lisc_commtable_create(LIS_MATRIX A)
{
	const LIS_INT P = A->nprocs, me = A->my_rank, n = A->n, ns = A->np - A->n;
	LIS_INT err = 0;
	LIS_INT *counts = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)P);
	if ((err = lisc_allgather_host(&ns, counts, sizeof(LIS_INT)))) { free(counts); return err; }
	LIS_INT maxns = 0;
	for (LIS_INT p = 0; p < P; p++) if (counts[p] > maxns) maxns = counts[p];
	LIS_INT *mine = (LIS_INT *)calloc((size_t)(maxns > 0 ? maxns : 1), sizeof(LIS_INT));
	LIS_INT *all = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(maxns > 0 ? maxns : 1) * (size_t)P);
	if (ns) memcpy(mine, A->l2g_map, sizeof(LIS_INT) * (size_t)ns);
	if ((err = lisc_allgather_host(mine, all, sizeof(LIS_INT) * (size_t)(maxns > 0 ? maxns : 1)))) { free(counts); free(mine); free(all); return err; }
	const size_t stride = (size_t)(maxns > 0 ? maxns : 1);

	LIS_INT *imcnt = (LIS_INT *)calloc((size_t)P, sizeof(LIS_INT)), *excnt = (LIS_INT *)calloc((size_t)P, sizeof(LIS_INT));
	for (LIS_INT i = 0, k = 0; i < ns; i++) { while (A->l2g_map[i] >= A->ranges[k + 1]) k++; imcnt[k]++; }
	for (LIS_INT p = 0; p < P; p++) {
		if (p == me) continue;
		const LIS_INT *l = all + (size_t)p * stride;
		for (LIS_INT i = 0; i < counts[p]; i++) if (l[i] >= A->is && l[i] < A->ie) excnt[p]++;
	}
	LIS_INT nb = 0, extot = 0;
	for (LIS_INT p = 0; p < P; p++) if (imcnt[p] || excnt[p]) { nb++; extot += excnt[p]; }

	LIS_COMMTABLE t = (LIS_COMMTABLE)calloc(1, sizeof(struct LIS_COMMTABLE_STRUCT));
	t->neibpe = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb > 0 ? nb : 1));
	t->import_ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb + 1));
	t->export_ptr = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(nb + 1));
	t->import_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(ns > 0 ? ns : 1));
	t->export_index = (LIS_INT *)malloc(sizeof(LIS_INT) * (size_t)(extot > 0 ? extot : 1));
	t->import_ptr[0] = t->export_ptr[0] = 0;
	LIS_INT q = 0;
	for (LIS_INT p = 0; p < P; p++) {
		if (!imcnt[p] && !excnt[p]) continue;
		t->neibpe[q] = p;
		t->import_ptr[q + 1] = t->import_ptr[q] + imcnt[p];
		t->export_ptr[q + 1] = t->export_ptr[q] + excnt[p];
		const LIS_INT *l = all + (size_t)p * stride;
		LIS_INT w = t->export_ptr[q];
		for (LIS_INT i = 0; i < counts[p]; i++) if (l[i] >= A->is && l[i] < A->ie) t->export_index[w++] = l[i] - A->is;
		q++;
	}
	for (LIS_INT i = 0; i < ns; i++) t->import_index[i] = n + i;    /* ghosts land where g2l numbered them */
	t->comm = A->comm; t->pad = A->pad_comm;
	t->neibpetot = nb; t->imnnz = ns; t->exnnz = extot;
	t->wssize = extot; t->wrsize = ns;
	t->ws = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(extot > 0 ? extot : 1));
	t->wr = (LIS_SCALAR *)malloc(sizeof(LIS_SCALAR) * (size_t)(ns > 0 ? ns : 1));
	A->commtable = t;
	free(counts); free(mine); free(all); free(imcnt); free(excnt);
	return LIS_SUCCESS;
}

/* ------------------------------------------------------------------ halo exchange on HBM pointers */
/* A neighbour whose export list is a run of consecutive rows -- a whole boundary plane of a structured grid in the row-block partition --
 * is sent STRAIGHT from x: no pack kernel, no staging copy.  export_run[i] = first row of the run, or -1 (irregular lists are packed
 * by liship_gather_f64 into d->ws as lis_send_recv packs them into ws, lis_matrix_mpi.c:866-874).  When every list is a run the pack
 * kernel is not launched at all. */
static LIS_INT halo_tables_ready(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	if (d->halo_ready) return LIS_SUCCESS;
	LIS_COMMTABLE t = A->commtable;
	d->all_runs = 1;
	d->export_run = (int *)malloc(sizeof(int) * (size_t)(t->neibpetot > 0 ? t->neibpetot : 1));
	if (!d->export_run) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", t->neibpetot);
	for (LIS_INT i = 0; i < t->neibpetot; i++) {
		const LIS_INT b = t->export_ptr[i], e = t->export_ptr[i + 1];
		int run = e > b ? t->export_index[b] : 0;
		for (LIS_INT k = b + 1; k < e && run >= 0; k++) if (t->export_index[k] != t->export_index[b] + (k - b)) run = -1;
		if (lisg.no_direct_halo) run = -1;
		d->export_run[i] = run;
		if (run < 0) d->all_runs = 0;
	}
	if (t->exnnz > 0) {
		HIPCHK(lisd_malloc((void **)&d->export_index, sizeof(int) * (size_t)t->exnnz));
		HIPCHK(liship_memcpy_h2d(d->export_index, t->export_index, sizeof(int) * (size_t)t->exnnz, lisg.stream));
		HIPCHK(lisd_malloc((void **)&d->ws, sizeof(double) * (size_t)t->exnnz));
		HIPCHK(liship_stream_synchronize(lisg.stream));
	}
	d->halo_ready = 1;
	return LIS_SUCCESS;
}

/* A solve that iterates in the numbering of a reordered plan (lis_solver.c) on several ranks: the rows a neighbour is sent are the same rows under their new
 * numbers -- export_index'[k] = position of row export_index[k] --, no list is a run any more (the pack kernel gathers them), the ghost slots x[n .. np) and
 * everything the neighbours see stay as they are: a rank renumbers (or not) on its own.  inner_rows: rows [0, inner_rows) of P A P^T read no ghost column
 * (liship_csr_plan_reordered_inner_rows): they run under the exchange as the interior planes of a slab do. */
LIS_INT lisc_halo_renumbered(LIS_MATRIX A, const int *perm, int inner_rows)
{
	lisd_mat *d = MDEV(A);
	LIS_COMMTABLE t = A->commtable;
	if (d->held_halo || !t) return LIS_SUCCESS;
	LISCHK(halo_tables_ready(A));
	int *ri = NULL, *rr = (int *)malloc(sizeof(int) * (size_t)(t->neibpetot > 0 ? t->neibpetot : 1));
	if (!rr) return LISI_ERR(LIS_ERR_OUT_OF_MEMORY, "malloc size = %D\n", t->neibpetot);
	for (LIS_INT i = 0; i < t->neibpetot; i++) rr[i] = -1;
	if (t->exnnz > 0) {
		LIS_INT err = lisd_malloc((void **)&ri, sizeof(int) * (size_t)t->exnnz);
		if (err) { free(rr); return err; }
		int rc = liship_permute_rows_of_list(A->n, perm, t->exnnz, d->export_index, ri, lisg.stream);
		if (rc) { (void)liship_free(ri); free(rr); return lisi_hip_error(__FILE__, __func__, __LINE__, rc); }
	}
	d->held_export_index = d->export_index; d->held_export_run = d->export_run; d->held_all_runs = d->all_runs;
	d->held_inner_begin = d->inner_begin; d->held_inner_end = d->inner_end;
	d->export_index = ri; d->export_run = rr; d->all_runs = t->exnnz > 0 ? 0 : 1;
	d->inner_begin = 0; d->inner_end = inner_rows < 0 ? 0 : inner_rows > A->n ? A->n : inner_rows;
	d->held_halo = 1;
	return LIS_SUCCESS;
}
void lisc_halo_restore(LIS_MATRIX A)
{
	lisd_mat *d = MDEV(A);
	if (!d->held_halo) return;
	(void)liship_free(d->export_index); free(d->export_run);
	d->export_index = d->held_export_index; d->export_run = d->held_export_run; d->all_runs = d->held_all_runs;
	d->inner_begin = d->held_inner_begin; d->inner_end = d->held_inner_end;
	d->held_export_index = NULL; d->held_export_run = NULL; d->held_halo = 0;
}

/* what a neighbour is sent: its run inside x, or its packed slice */
static const double *send_ptr(const lisd_mat *d, LIS_COMMTABLE t, LIS_INT i, const double *dx)
{
	return d->export_run[i] >= 0 ? dx + d->export_run[i] : d->ws + t->export_ptr[i];
}

static LIS_INT halo_pack(LIS_MATRIX A, double *dx)
{
	lisd_mat *d = MDEV(A);
	LIS_COMMTABLE t = A->commtable;
	if (t->exnnz > 0 && !d->all_runs) HIPCHK(liship_gather_f64(t->exnnz, d->export_index, dx, d->ws, lisg.stream));
	return LIS_SUCCESS;
}

static LIS_INT halo_rccl(LIS_MATRIX A, double *dx, void *stream)
{
	lisd_mat *d = MDEV(A);
	LIS_COMMTABLE t = A->commtable;
	const LIS_INT n = A->n, pad = t->pad;
	/* EVERY forward exchange goes over the halo communicator when there is one, whichever stream it is queued on.  Whether a product overlaps its exchange with the
	 * interior rows is decided rank by rank (lisd_spmv: "at least half of my rows touch no ghost column"), so two neighbours may well decide differently -- 3 planes
	 * per rank on 3 ranks: the end ranks overlap, the middle one does not -- and a Send on one communicator never meets a Recv on the other (the job hung
	 * until the watchdog).  The communicator must therefore not depend on that decision; only the STREAM does.  Order on each communicator is program order on every
	 * rank (exchanges on nccl_halo, folds and reverse exchanges on nccl_comm), and on one rank the two never run concurrently: an overlapped exchange is fenced
	 * by ev_packed / ev_landed between the folds before and after it (lisc_halo_begin / lisc_halo_end), a plain one sits in the library's stream with them.
	 * Keep it that way -- concurrent collectives on two communicators of one process are a known deadlock shape. */
	nccl_comm comm = lisg.nccl_halo ? lisg.nccl_halo : lisg.nccl_comm;
	NCCLCHK(rccl.GroupStart());
	for (LIS_INT i = 0; i < t->neibpetot; i++) {
		const LIS_INT peer = t->neibpe[i];
		const LIS_INT sc = t->export_ptr[i + 1] - t->export_ptr[i], rc = t->import_ptr[i + 1] - t->import_ptr[i];
		if (sc > 0) NCCLCHK(rccl.Send(send_ptr(d, t, i, dx), (size_t)sc, NCCL_DOUBLE, peer, comm, stream));
		if (rc > 0) NCCLCHK(rccl.Recv(dx + n + pad + t->import_ptr[i], (size_t)rc, NCCL_DOUBLE, peer, comm, stream));
	}
	NCCLCHK(rccl.GroupEnd());
	return LIS_SUCCESS;
}

/* host round trip (tests / bring-up only): the slices go to the host, through the callback, and back into x[n + pad ...) */
static LIS_INT halo_callbacks(LIS_MATRIX A, double *dx)
{
	lisd_mat *d = MDEV(A);
	LIS_COMMTABLE t = A->commtable;
	const LIS_INT n = A->n, pad = t->pad;
	for (LIS_INT i = 0; i < t->neibpetot; i++) {
		const LIS_INT sc = t->export_ptr[i + 1] - t->export_ptr[i];
		if (sc > 0) HIPCHK(liship_memcpy_d2h(t->ws + t->export_ptr[i], send_ptr(d, t, i, dx), sizeof(double) * (size_t)sc, lisg.stream));
	}
	HIPCHK(liship_stream_synchronize(lisg.stream));
	if (lisg.cb.neighbor_exchange(lisg.cb.ctx, t->neibpetot, t->neibpe, t->ws, t->export_ptr, t->wr, t->import_ptr))
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "neighbor_exchange callback failed\n");
	if (t->imnnz > 0) HIPCHK(liship_memcpy_h2d(dx + n + pad, t->wr, sizeof(double) * (size_t)t->imnnz, lisg.stream));
	return LIS_SUCCESS;
}

LIS_INT lisc_halo_device(LIS_MATRIX A, double *dx)
{
	LIS_COMMTABLE t = A->commtable;
	if (!t || t->neibpetot == 0) return LIS_SUCCESS;
	LISCHK(halo_tables_ready(A));
	LISCHK(halo_pack(A, dx));
	if (lisg.comm_kind == 1) return halo_rccl(A, dx, lisg.stream);
	if (lisg.comm_kind == 2) return halo_callbacks(A, dx);
	return LISI_ERR(LIS_ERR_ILL_ARG, "halo exchange without a communicator\n");
}

/* the same exchange split in two so the rows that reference no ghost column run while the planes travel:
 *   begin: pack on the compute stream (nothing when every list is a run); RCCL send/recv on the second stream behind an event
 *          that marks "x is complete" on the compute stream
 *   end:   the compute stream waits for the landing (stream-side wait, the host does not block)
 * With the callback backend (tests) `begin` only packs and `end` does the host round trip. */
LIS_INT lisc_halo_begin(LIS_MATRIX A, double *dx)
{
	LIS_COMMTABLE t = A->commtable;
	if (!t || t->neibpetot == 0) return LIS_SUCCESS;
	LISCHK(halo_tables_ready(A));
	LISCHK(halo_pack(A, dx));
	if (lisg.comm_kind != 1) return LIS_SUCCESS;
	if (!lisg.comm_stream) {
		HIPCHK(liship_stream_create(&lisg.comm_stream));
		HIPCHK(liship_event_create(&lisg.ev_packed));
		HIPCHK(liship_event_create(&lisg.ev_landed));
	}
	HIPCHK(liship_event_record(lisg.ev_packed, lisg.stream));          /* x (and the packed slices) are final behind this point */
	HIPCHK(liship_stream_wait_event(lisg.comm_stream, lisg.ev_packed));
	LISCHK(halo_rccl(A, dx, lisg.comm_stream));
	HIPCHK(liship_event_record(lisg.ev_landed, lisg.comm_stream));
	return LIS_SUCCESS;
}

LIS_INT lisc_halo_end(LIS_MATRIX A, double *dx)
{
	LIS_COMMTABLE t = A->commtable;
	if (!t || t->neibpetot == 0) return LIS_SUCCESS;
	if (lisg.comm_kind == 1) { HIPCHK(liship_stream_wait_event(lisg.stream, lisg.ev_landed)); return LIS_SUCCESS; }
	if (lisg.comm_kind == 2) return halo_callbacks(A, dx);
	return LISI_ERR(LIS_ERR_ILL_ARG, "halo exchange without a communicator\n");
}

/* the reverse exchange of lis_reduce (ref lis_matrix_mpi.c:959-1000): every rank sends the ghost part of y to
 * the ghosts' owners, which add what they receive to their own rows, neighbour by neighbour in table order
 * (one scatter-add launch per neighbour: rows are unique within a neighbour's list, not across lists) */
LIS_INT lisc_reduce_device(LIS_MATRIX A, double *dy)
{
	LIS_COMMTABLE t = A->commtable;
	lisd_mat *d = MDEV(A);
	if (!t || t->neibpetot == 0) return LIS_SUCCESS;
	LISCHK(halo_tables_ready(A));
	const LIS_INT n = A->n, pad = t->pad;
	if (!d->wr) HIPCHK(lisd_malloc((void **)&d->wr, sizeof(double) * (size_t)(t->exnnz > 0 ? t->exnnz : 1)));
	if (lisg.comm_kind == 1) {
		NCCLCHK(rccl.GroupStart());
		for (LIS_INT i = 0; i < t->neibpetot; i++) {
			const LIS_INT peer = t->neibpe[i];
			const LIS_INT sc = t->import_ptr[i + 1] - t->import_ptr[i], rc = t->export_ptr[i + 1] - t->export_ptr[i];
			if (sc > 0) NCCLCHK(rccl.Send(dy + n + pad + t->import_ptr[i], (size_t)sc, NCCL_DOUBLE, peer, lisg.nccl_comm, lisg.stream));
			if (rc > 0) NCCLCHK(rccl.Recv(d->wr + t->export_ptr[i], (size_t)rc, NCCL_DOUBLE, peer, lisg.nccl_comm, lisg.stream));
		}
		NCCLCHK(rccl.GroupEnd());
	} else if (lisg.comm_kind == 2) {           /* host round trip (tests / bring-up only) */
		if (t->imnnz > 0) HIPCHK(liship_memcpy_d2h(t->wr, dy + n + pad, sizeof(double) * (size_t)t->imnnz, lisg.stream));
		HIPCHK(liship_stream_synchronize(lisg.stream));
		if (lisg.cb.neighbor_exchange(lisg.cb.ctx, t->neibpetot, t->neibpe, t->wr, t->import_ptr, t->ws, t->export_ptr))
			return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "neighbor_exchange callback failed\n");
		if (t->exnnz > 0) HIPCHK(liship_memcpy_h2d(d->wr, t->ws, sizeof(double) * (size_t)t->exnnz, lisg.stream));
	} else return LISI_ERR(LIS_ERR_ILL_ARG, "reduce without a communicator\n");
	for (LIS_INT i = 0; i < t->neibpetot; i++) {
		const LIS_INT rc = t->export_ptr[i + 1] - t->export_ptr[i];
		if (rc > 0) HIPCHK(liship_scatter_add_f64(rc, d->export_index + t->export_ptr[i], d->wr + t->export_ptr[i], dy, lisg.stream));
	}
	return LIS_SUCCESS;
}

/* host-array variant of the same exchange (the reference's lis_send_recv on x[]), used by CPU tests of the
 * tables: x has np entries, ghosts are filled in place */
LIS_INT lis_amd_halo_exchange_host(LIS_MATRIX A, LIS_SCALAR x[])
{
	LIS_COMMTABLE t = A->commtable;
	if (!t || t->neibpetot == 0) return LIS_SUCCESS;
	if (lisg.comm_kind != 2) return LISI_ERR(LIS_ERR_ILL_ARG, "host halo exchange needs the callback communicator\n");
	for (LIS_INT i = 0; i < t->exnnz; i++) t->ws[i] = x[t->export_index[i]];
	if (lisg.cb.neighbor_exchange(lisg.cb.ctx, t->neibpetot, t->neibpe, t->ws, t->export_ptr, t->wr, t->import_ptr))
		return LISI_ERR(LIS_ERR_NOT_IMPLEMENTED, "neighbor_exchange callback failed\n");
	for (LIS_INT i = 0; i < t->imnnz; i++) x[t->import_index[i] + t->pad] = t->wr[i];
	return LIS_SUCCESS;
}
