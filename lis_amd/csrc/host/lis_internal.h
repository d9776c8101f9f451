/*
 * lis_internal.h -- private side of liblis_amd.so's host layer (C).
 *
 * Public Lis objects are allocated as a larger private struct whose first member is the public,
 * ABI-identical struct from include/lis.h; the tail carries the HBM copy and its validity flags.
 */
#ifndef LIS_AMD_INTERNAL_H
#define LIS_AMD_INTERNAL_H

#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "lis.h"
#include "lis_amd.h"
#include "liship.h"

/* ---- error reporting: "file(line) : func : error CODE :message" (reference src/system/lis_error.c:161) */
LIS_INT lisi_error(const char *file, const char *func, int line, LIS_INT code, const char *fmt, ...);
#define LISI_ERR(code, ...) (lisi_error(__FILE__, __func__, __LINE__, (code), __VA_ARGS__), (code))
/* a failing HIP call is fatal for the operation: no CPU fallback exists */
LIS_INT lisi_hip_error(const char *file, const char *func, int line, int hipcode);
#define HIPCHK(expr) do { int rc__ = (expr); if (rc__ != 0) return lisi_hip_error(__FILE__, __func__, __LINE__, rc__); } while (0)
#define LISCHK(expr) do { LIS_INT e__ = (expr); if (e__ != LIS_SUCCESS) return e__; } while (0)

/* ---- object registry (lis_is_malloc on handles, leak-free lis_finalize) */
#define LISI_KIND_VECTOR 1
#define LISI_KIND_MATRIX 2
#define LISI_KIND_SOLVER 3
#define LISI_KIND_PRECON 4
void lisi_register(void *obj, int kind);
void lisi_unregister(void *obj);
int  lisi_is_registered(void *obj);

/* ---- vectors ------------------------------------------------------------------------------------ */
typedef struct {
	double *d;          /* HBM buffer (NULL until first device use) */
	size_t  cap;        /* doubles allocated in HBM */
	size_t  hlen;       /* doubles allocated on the host (value[]) */
	int     host_valid; /* value[] holds the current data */
	int     dev_valid;  /* d[] holds the current data */
	void   *region;     /* value[] lives on pages of its own whose protection follows the flags above (lis_pages.c); NULL: plain memory */
} lisd_vec;

typedef struct {
	struct LIS_VECTOR_STRUCT pub;
	lisd_vec dev;
} lisi_vector;
#define VDEV(v) (&((lisi_vector *)(v))->dev)

/* ---- matrices ----------------------------------------------------------------------------------- */
typedef struct {
	int ready;                 /* HBM copy is built */
	int host_written;          /* the program wrote one of the host arrays the copy was built from (page fault, lis_pages.c): rebuilt before the next use */
	int solve_holds;           /* lis_solve_kernel swapped the renumbered form in (plan, ptr, index, value are the plan's P A P^T): no rebuild, no free until it swaps back */
	int checked;               /* LIS_AMD_MATRIX_CHECK=1: host_hash holds the hash of the host arrays the copy was built from */
	unsigned long long host_hash;
	int device_only;           /* arrays were adopted from the caller; no host copy exists */
	long long served;          /* products this HBM copy has served (lisd_spmv and the fused product + dot): the lazy renumbering waits for lisg.reorder_after of them */
	int reorder_tried;         /* the renumbered form was built, or found not worth building, for this copy */
	int type;                  /* kernel family actually used (CSC is served as transposed CSR) */
	int n, np, nnz;
	int maxnzr, nnd, nr, nc, bnr, bnc;
	int *ptr, *index, *row, *bptr, *bindex;
	double *value;
	liship_csr_plan_t plan;
	int xs_rows;               /* native ELL / DIA: rows per plane of the structured grid (largest offset most rows reach), 0 = none: the XCD strips of their kernels (lis_device.c fmt_strips) */
	unsigned char *ell_codes;  /* ELL: one-byte column codes + dictionary when the matrix allows it (liship_ell_encode_indices) */
	int *ell_dict;
	/* A^T as a CSR in the reference's scatter order (lis_matvech.c), built on the first lis_matvech */
	int t_ready, t_rows, t_nnz;
	int *t_ptr, *t_index;
	double *t_value;
	liship_csr_plan_t t_plan;
	double *wr;                /* received ghost contributions (reverse halo) */
	/* the same for the renumbered form (P A P^T)^T: swapped with the set above while a solve iterates in the plan's numbering (lis_solver.c), so that BiCG & co.
	 * find a transposed copy in the numbering they run in */
	int rt_ready, rt_nnz;
	int *rt_ptr, *rt_index;
	double *rt_value;
	liship_csr_plan_t rt_plan;
	double *t_diag;            /* split CSR: the diagonal, added to the off-diagonal sums of A^T x by one element-wise pass (lis_matvech.c) */
	/* halo (multi-GPU) */
	int halo_ready;
	int *export_index;         /* device copy of commtable->export_index */
	int *export_run;           /* host, per neighbour: the first row when its export list is a run of consecutive rows (sent straight from x), else -1 */
	int all_runs;              /* every neighbour's list is a run: no pack kernel at all (whole boundary planes of a structured grid) */
	double *ws;                /* packed send buffer in HBM */
	int inner_begin, inner_end;/* maximal run of rows without ghost columns: overlappable with the halo */
	/* while a multi-rank solve iterates in the numbering of a reordered plan (lisc_halo_renumbered): the tables above hold the renumbered export list, these the caller's */
	int held_halo, held_all_runs, held_inner_begin, held_inner_end;
	int *held_export_index, *held_export_run;
	/* scratch for the raw-array entry points lis_matvec_<fmt>(A, x[], y[]) */
	double *sx, *sy; size_t scap;
	/* split JAD matrix (lis_split.c): d->ptr/index/value/plan hold L, these hold U, the diagonal and a work vector */
	int split_jad;
	int *u_ptr, *u_index;
	double *u_value, *dsplit, *jw;
	liship_csr_plan_t u_plan;
} lisd_mat;

typedef struct {
	struct LIS_MATRIX_STRUCT pub;
	lisd_mat dev;
} lisi_matrix;
#define MDEV(A) (&((lisi_matrix *)(A))->dev)

/* ---- global runtime state ------------------------------------------------------------------------ */
typedef struct {
	int initialized;
	int residency;
	int no_fusion;             /* LIS_AMD_NO_FUSION=1: Krylov loops call the separate kernels (A/B measurements) */
	int device_ready;
	int device;
	void *stream;
	void *reduce_work;         /* HBM scratch of the two-stage reductions */
	double *reduce_out;        /* HBM: up to 4 results */
	double *gather_out;        /* HBM: nprocs * 4 doubles (cross-rank fold) */
	double *host_out;          /* page-locked: 4 results, and up to 4 x 64 gathered partials */
	/* communicator */
	int rank, nprocs;
	int comm_kind;             /* 0 none, 1 rccl, 2 callbacks */
	void *nccl_comm;
	void *nccl_halo;           /* a second communicator for the halo exchange on the second stream (NULL: the first one serves, ordered by events) */
	void *comm_stream;         /* second HIP stream: halo send/recv overlapped with the interior rows */
	void *ev_packed, *ev_landed;
	int reference_layout;      /* LIS_AMD_REFERENCE_LAYOUT=1 / lis_amd_set_reference_layout(1): products stream the reference's own arrays (lis_device.c) */
	int long_row_chain;        /* LIS_AMD_LONG_ROW_CHAIN=1: the part of a row beyond the LDS stage is ONE left-to-right chain (the reference's bits) instead of the default workgroup tree */
	int no_value_records;      /* LIS_AMD_NO_VALUE_RECORDS=1: matrices whose rows repeat with their values keep streaming the values (A/B measurements) */
	int eager_coherence;       /* LIS_AMD_COHERENCE=eager / lis_amd_set_coherence(0): COHERENT copies on every call instead of following page faults (lis_pages.c) */
	int no_device_convert;     /* LIS_AMD_NO_DEVICE_CONVERT=1: lis_matrix_convert always works on the host arrays (A/B, tests of the host routines) */
	int no_row_form;           /* LIS_AMD_NO_ROW_FORM=1 / lis_amd_set_row_form(0): constant-coefficient ELL / DIA matrices keep their native layout and kernels */
	int no_local_short_rows;   /* LIS_AMD_NO_LOCAL_SHORT_ROWS=1: plans of short rows never try block-local columns (the rule of rounds 2-5; A/B) */
	int no_row_patterns;       /* LIS_AMD_NO_ROW_PATTERNS=1: coded CSR matrices keep one byte per non-zero instead of one per row (A/B measurements) */
	int last_uniform_jacobi;   /* the last lis_solve ran CG + Jacobi with 1/diag as one double (lis_amd_last_solve_uniform_jacobi) */
	int graphs;                /* LIS_AMD_GRAPHS=1: single-rank device-driven loops replay a hipGraph of one batch (opt-in: measured, no gain) */
	int last_renumbered;       /* the last lis_solve ran in the numbering of a reordered plan (lis_amd_last_solve_renumbered) */
	int last_graph_replays;    /* batches of the last lis_solve that were graph replays (lis_amd_last_solve_graph_replays) */
	int no_uniform_jacobi;     /* LIS_AMD_NO_UNIFORM_JACOBI=1: CG + Jacobi reads 1/diag even when the diagonal is constant (A/B measurements) */
	int row_block_dots;        /* LIS_AMD_ROW_BLOCK_DOTS=1: fused dots of the dominant-pattern product as the row blocks' partial sums */
	int no_marching;           /* LIS_AMD_NO_MARCHING=1: 7-point matrices with value records keep the gathering dominant-pattern kernel (round 3's headline kernel: A/B measurements) */
	int no_team_kernels;       /* LIS_AMD_NO_TEAM_KERNELS=1: patterned rows of 8..32 entries and long BSR block rows keep the round-2 kernels (A/B measurements) */
	double last_input_s, last_input_assemble_s;      /* lis_amd_last_input_times */
	long long reorder_after;   /* LIS_AMD_REORDER_AFTER=K: the renumbered form of a plan is built by the first lis_solve that finds K products served (default 4096; 0: at plan time) */
	int plain_malloc;          /* LIS_AMD_PLAIN_MALLOC=1 / lis_amd_set_matrix_pages(0): lis_matrix_malloc_<fmt> returns malloc memory (free()-compatible, never protected, edits not seen) */
	int no_reorder;            /* LIS_AMD_NO_REORDER=1: long-row CSR plans keep the caller's numbering whatever their lists look like (A/B measurements) */
	int no_local_columns;      /* LIS_AMD_NO_LOCAL_COLUMNS=1: long-row CSR products keep the 4 B column indices (A/B measurements) */
	int no_index_codes;        /* LIS_AMD_NO_INDEX_CODES=1: CSR products keep reading the 4 B column indices (A/B measurements) */
	int host_scalars;          /* LIS_AMD_HOST_SCALARS=1: CG / BiCGSTAB read every scalar back (A/B against the device-driven loops) */
	int no_overlap;            /* LIS_AMD_NO_OVERLAP=1: exchange first, then the whole product (A/B measurements) */
	int ref_reductions;        /* LIS_AMD_REFERENCE_REDUCTIONS=T / lis_amd_set_reference_reductions(T): sums in the reference's order for T threads (parity mode) */
	int matrix_check;          /* LIS_AMD_MATRIX_CHECK=1 / lis_amd_set_matrix_check(1): every use of a matrix re-hashes its host arrays and rebuilds a stale HBM copy (debugging aid for caller-malloc'ed arrays) */
	int no_direct_halo;        /* LIS_AMD_NO_DIRECT_HALO=1: boundary rows that form a run are packed like any other list instead of being sent straight from x (A/B) */
	lis_amd_comm_callbacks cb;
} lisi_globals;
extern lisi_globals lisg;

/* ---- page-protected host arrays (lis_pages.c) */
#define LISP_RW   0      /* host array holds the data, HBM copy stale (or none) */
#define LISP_RO   1      /* both agree: a host write faults */
#define LISP_NONE 2      /* HBM copy holds the data: any host access faults */
LIS_SCALAR *lisp_alloc(LIS_VECTOR v, size_t doubles);
void lisp_free(LIS_VECTOR v);
LIS_INT lisp_grow(LIS_VECTOR v, size_t doubles);
void lisp_protect(LIS_VECTOR v, int prot);
int  lisp_state(LIS_VECTOR v);
LIS_INT lisp_vec_home(LIS_VECTOR v);                          /* value[] current on the host, the copy written through the alias mapping (lazy coherence) */
int  lisp_lazy(void);
void lisp_check_handler(void);                                    /* is the SIGSEGV disposition still the library's?  (else: eager coherence from here on, with a message) */
void *lisp_alloc_lazy(void *matrix, size_t bytes_used, void *dev, int own_dev);   /* a matrix array held in HBM until its first host touch */
int  lisp_free_array(void *p);                                /* 1: p lived on such pages (unmapped), 0: plain memory (caller frees) */
LIS_INT lisp_fill_matrix(void *matrix);
int  lisp_lazy_arrays(void *matrix);
void lisp_reown(void *from, void *to);
void *lisp_alloc_tracked(size_t bytes_used);                  /* an array of lis_matrix_malloc_<fmt>: pages whose writes are seen once a matrix adopted and uploaded it */
int  lisp_adopt(void *matrix, void *array);
int  lisp_matrix_protect(void *matrix);
void lisp_matrix_release(void *matrix, int forget);
int  lisp_protected_arrays(void *matrix);
size_t lisp_array_bytes(void *p);
LIS_INT lisd_staged_d2h(void *dst, const void *src, size_t bytes);   /* HBM -> pageable host memory through the pinned staging buffers */
LIS_INT lisd_staged_d2h_fault(void *dst, const void *src, size_t bytes);   /* the same from inside the page-fault handler: a stream and buffers of its own, no OpenMP */
LIS_INT lisd_fault_stage_prepare(void);                             /* (made when the handler is installed, never inside it) */
void    lisd_capture_mark(int on);                                  /* a hipGraph capture of the library's stream begins / ends: fault-time copies wait it out */
LIS_INT lisd_vec_host_write(LIS_VECTOR v, int keep);          /* the library is about to write value[] on the host (keep: current data needed first) */

/* ---- device runtime (lis_device.c) */
LIS_INT lisd_init(void);                                     /* lazy; fails loudly without a GPU */
LIS_INT lisd_vec_in(LIS_VECTOR v, double **d);                /* device pointer holding current data (input) */
LIS_INT lisd_vec_out(LIS_VECTOR v, double **d);               /* device pointer to be fully overwritten (n entries) */
LIS_INT lisd_vec_done(LIS_VECTOR v);                          /* after a kernel wrote v on the device */
LIS_INT lisd_vec_reserve(LIS_VECTOR v, size_t doubles);       /* grow the HBM buffer (keeps data) */
LIS_INT lisd_vec_to_host(LIS_VECTOR v);
void    lisd_vec_free(LIS_VECTOR v);
LIS_INT lisd_mat_ready(LIS_MATRIX A);
void    lisd_mat_free(LIS_MATRIX A);
LIS_INT lisd_init_quiet(void);                               /* lisd_init without the diagnostic when no device exists */
void    lisd_mat_eager(LIS_MATRIX A);                         /* resident mode: upload at assemble / convert time */
LIS_INT lisd_pool_get(size_t bytes, void **out);              /* HBM buffer of exactly `bytes`, reused across solves */
void    lisd_pool_put(void *p, size_t bytes);
int     lisd_malloc(void **out, size_t bytes);                /* liship_malloc that hands the pool back and retries when HBM is full; HIP code */
int     lis_amd_trim_count(void);                             /* lis_amd_trim(), returning the number of buffers released */
LIS_INT lisd_mat_ready_t(LIS_MATRIX A);                       /* build / upload the transposed operator */
LIS_INT lisd_spmv_t(LIS_MATRIX A, double *dx, double *dy);    /* y[0..np) = A^T x, ghost rows reduced to owners */
LIS_INT lisd_spmv(LIS_MATRIX A, double *dx, double *dy);      /* y = A x on device pointers (halo included) */
LIS_INT lisd_csr_plan(liship_csr_plan_t *plan, int n, const int *dptr, const int *dindex, const double *dvalue);   /* row split + index codes */
LIS_INT lisd_csr_plan_cols(liship_csr_plan_t *plan, int n, int ncols, const int *dptr, const int *dindex, const double *dvalue);   /* ... of a rank's local rows with ghost columns [n, ncols) */
LIS_INT lisd_csr_plan_plain(liship_csr_plan_t *plan, int n, const int *dptr, const int *dindex, const double *dvalue);    /* the same without a renumbered form (matrices no solve iterates on) */
LIS_INT lisd_spmv_dot_launch(LIS_MATRIX A, double *dx, double *dy, const double *dw, int want_sumsq); /* sums -> reduce_out */
LIS_INT lisd_spmv_dot_launch_to(LIS_MATRIX A, double *dx, double *dy, const double *dw, int want_sumsq, double *result); /* sums -> result (HBM) */
LIS_INT lisd_fetch(int count, double *out);                   /* reduce_out[0..count) -> host, cross-rank fold */
LIS_INT lisd_dot(int n, const double *dx, const double *dy, double *out);
LIS_INT lisd_nrm2(int n, const double *dx, double *out);
LIS_INT lisd_nrm1(int n, const double *dx, double *out);
LIS_INT lisd_dot2(int n, const double *dx, const double *dy, double *out2);

/* ---- communicator (lis_comm.c) */
LIS_INT lisi_matrix_retype(LIS_MATRIX A, LIS_INT want, LIS_INT block);       /* convert an assembled matrix in place (lis_io.c) */
LIS_INT lisc_ranges_create(LIS_Comm comm, LIS_INT *local_n, LIS_INT *global_n, LIS_INT **ranges,
                           LIS_INT *is, LIS_INT *ie, LIS_INT *nprocs, LIS_INT *my_rank);
LIS_INT lisc_matrix_g2l(LIS_MATRIX A);                        /* global -> local columns, ghosts appended */
LIS_INT lisc_commtable_create(LIS_MATRIX A);
void    lisc_commtable_destroy(LIS_COMMTABLE t);
LIS_INT lisc_reduce_device(LIS_MATRIX A, double *dy);         /* dy[export rows] += neighbours' dy[n..np) */
LIS_INT lisd_mat_lazy_reorder(LIS_MATRIX A);                  /* the renumbered form of A's plan once it has served lisg.reorder_after products (lis_solve) */
LIS_INT lisc_halo_begin(LIS_MATRIX A, double *dx);            /* pack + start the exchange (second stream) */
LIS_INT lisc_halo_end(LIS_MATRIX A, double *dx);              /* ghosts of dx are valid for work queued after this */
LIS_INT lisc_halo_device(LIS_MATRIX A, double *dx);           /* fill dx[n..np) from the neighbours */
LIS_INT lisc_halo_renumbered(LIS_MATRIX A, const int *perm, int inner_rows);   /* the export list in a reordered plan's numbering (perm[new position] = row, device) swapped in ... */
void    lisc_halo_restore(LIS_MATRIX A);                      /* ... and the caller's back */
LIS_INT lisc_gather_device(const double *src, int count);    /* RCCL only: every rank's src[0..count) -> lisg.gather_out, rank-major, on the stream */
LIS_INT lisc_fold(int count, double *host_inout);             /* sum over ranks, rank order */
LIS_INT lisc_allgather_host(const void *send, void *recv, size_t bytes);

/* ---- matrix internals shared between files */
LIS_INT lisi_matrix_check(LIS_MATRIX A, int level);
int     lisi_host_threads(void);      /* threads the host-side conversions may use: affinity mask capped by the cgroup CPU quota, at most 32 */
/* ---- split form (lis_split.c) */
void    lisi_sortr_ii(LIS_INT lo, LIS_INT hi, LIS_INT *key, LIS_INT *tag);    /* the reference's descending quicksort (lis_convert.c) */
void    lisi_matrix_dlu_destroy(LIS_MATRIX A);
LIS_INT lisi_split_rows(LIS_MATRIX A, LIS_INT *rows, LIS_INT **ptr, LIS_INT **idx, LIS_SCALAR **val, int *from_zero);
LIS_INT lisi_matrix_bscale_bsr(LIS_MATRIX A, LIS_VECTOR B);                  /* -scale jacobi -storage bsr (lis_scale.c) */
LIS_INT lisi_split_jad_part(LIS_MATRIX A, int upper, LIS_INT **ptr, LIS_INT **idx, LIS_SCALAR **val);
#define LISI_CHECK_NULL 0
#define LISI_CHECK_SIZE 1
#define LISI_CHECK_ASSEMBLED 2
#define LISI_CHECK_NOT_ASSEMBLED 3
LIS_INT lisi_matrix_storage_destroy(LIS_MATRIX A);
LIS_INT lisi_matrix_copy_header(LIS_MATRIX src, LIS_MATRIX dst);
void    lisi_sort_row(LIS_INT lo, LIS_INT hi, LIS_INT *idx, LIS_SCALAR *val);
LIS_INT lisi_convert_csr_to(LIS_MATRIX Ain, LIS_MATRIX Aout);  /* Aout->matrix_type selects the target */
LIS_INT lisi_convert_to_csr(LIS_MATRIX Ain, LIS_MATRIX Aout);
LIS_INT lisi_jad_order(LIS_MATRIX A, LIS_INT *maxnzr, LIS_INT **perm, LIS_INT **ptr);      /* the reference's length-sorted row order + jagged-diagonal starts */
LIS_INT lisd_convert_csr(LIS_MATRIX Ain, LIS_MATRIX Aout, int *done);   /* csr -> ell / dia / csc / bsr in HBM when Ain lives there (lis_device.c) */
LIS_INT lisi_matrix_deep_copy(LIS_MATRIX Ain, LIS_MATRIX Aout);

/* args (lis_initialize / lis_solver_set_option share the tokenizer) */
int lisi_tokenize(const char *text, char ***tokens);
void lisi_tokens_free(char **tokens, int n);
extern int lisi_cmd_argc; extern char **lisi_cmd_argv;

#endif
