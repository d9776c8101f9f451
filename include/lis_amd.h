/*
 * lis_amd.h -- extensions of liblis_amd.so beyond the Lis API: where the data lives, and multi-GPU.
 *
 * Lis objects expose raw host arrays (v->value[], A->ptr[] ...) that callers read and write directly
 * (test/spmvtest1.c:215, src/solver/lis_solver_gmres.c:203).  The kernels run on HBM copies.  Two
 * residency policies decide who is authoritative:
 *
 *   LIS_AMD_COHERENT (default)  any program written for Lis works unchanged: what it reads from v->value[] is what the last call
 *       left there, what it writes there is what the next call uses.  Vectors are always library-allocated (lis_vector_create /
 *       lis_vector_duplicate), so value[] lives on pages of its own whose PROTECTION follows the HBM copy (lis_pages.c): a kernel that
 *       writes a vector leaves its pages inaccessible and copies nothing; the first host access faults, the handler brings the vector
 *       home; a host WRITE to a vector both sides agree on faults once and marks the HBM copy stale.  Call sequences that stay inside
 *       the API therefore run at the speed of LIS_AMD_RESIDENT, and only the vectors a program really touches cross PCIe.
 *       LIS_AMD_COHERENCE=eager / lis_amd_set_coherence(0) selects the older implementation of the same semantics -- every call
 *       uploads its inputs and downloads its outputs -- for programs that pass v->value to something a page fault cannot interrupt
 *       (write(2) and other system calls fail with EFAULT on a protected buffer; another device's DMA).  Matrices keep their rule: the
 *       host arrays are the truth, the HBM copy is built once, lis_amd_matrix_host_modified() after changing them.
 *   LIS_AMD_RESIDENT            objects live in HBM; validity of each side is tracked by the library.
 *       API writers (lis_vector_set_value(s), set_all ...) and readers (get_value(s), gather) stay
 *       correct; code that pokes v->value[] directly must bracket it with
 *       lis_amd_vector_sync_host() (before reading) / lis_amd_vector_host_modified() (after writing).
 */
#ifndef LIS_AMD_H
#define LIS_AMD_H

#include "lis.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A HIP runtime or RCCL failure (a kernel launch, a copy, a stream synchronize, a collective) is reported as LIS_AMD_ERR_DEVICE with the HIP error string on stderr;
 * hipErrorOutOfMemory as LIS_ERR_OUT_OF_MEMORY.  (The reference has no device and no such code: 7 is the first value lis.h:1052-1063 leaves free.  It is NOT
 * LIS_ERR_NOT_IMPLEMENTED, which keeps its meaning: a storage format, solver or preconditioner this library does not serve.) */
#define LIS_AMD_ERR_DEVICE 7

#define LIS_AMD_COHERENT 0
#define LIS_AMD_RESIDENT 1

LIS_INT lis_amd_set_residency(LIS_INT mode);
LIS_INT lis_amd_get_residency(void);
/* COHERENT by page protection (1, default) or by copies on every call (0); see above */
LIS_INT lis_amd_set_coherence(LIS_INT lazy);
/* test / diagnostic hooks: protection of v->value's pages (0 read + write: host holds the data; 1 read-only: both agree; 2 none: the HBM
 * copy holds it; -1: plain memory), and how many read / write faults the handler has served */
/* arrays of A (a matrix made by lis_matrix_convert in HBM) that no host access has asked for yet: they exist in HBM only (tests) */
LIS_INT lis_amd_matrix_lazy_arrays(LIS_MATRIX A);
/* lis_matrix_convert of a CSR matrix that lives in HBM builds ELL / DIA / CSC / BSR there (kernels/convert.hip: the reference's arrays, bit for
 * bit) and leaves the new matrix's host arrays to their first touch (1, default); 0 (env LIS_AMD_NO_DEVICE_CONVERT=1): always on the host arrays */
LIS_INT lis_amd_set_device_convert(LIS_INT on);
LIS_INT lis_amd_vector_page_state(LIS_VECTOR v);
LIS_INT lis_amd_vector_page_protect(LIS_VECTOR v, LIS_INT state);      /* force a protection (tests of the fault handler without a GPU) */
LIS_INT lis_amd_page_faults(LIS_INT *reads, LIS_INT *writes);
/* lazy coherence lives on the process's SIGSEGV disposition (one line on stderr says so when it is taken; LIS_AMD_QUIET=1 drops the line).  A handler the program
 * installs LATER would take the protected pages' faults away: the library checks at every lis_solve (and now and then in between), and when the disposition is no
 * longer its own it brings home what only HBM holds, opens every page for good and goes on in eager coherence, with a line on stderr.  This call runs the check
 * now; 1: the library's handler is in place, 0: it is not (never installed, or replaced) */
LIS_INT lis_amd_check_fault_handler(void);
LIS_INT lis_amd_page_fault_waits(void);                /* faults that found another thread bringing the same array home and waited for its copy */
/* tests of the handler without a GPU: the host buffer `src` (n + pad doubles) plays the HBM copy of v -- v's pages lose all access and the first
 * touch copies src home in two halves delay_ms apart (delay_ms < 0: the second half is held until -delay_ms other threads wait for the copy, 10 s at most),
 * through the library's alias mapping (lis_pages.c); src = NULL removes the hook */
LIS_INT lis_amd_vector_page_test_source(LIS_VECTOR v, const LIS_SCALAR *src, LIS_INT delay_ms);

/* How the CG / BiCGSTAB loops run (all three produce identical bits; the choice is for A/B measurements):
 *   DEVICE  scalars live in HBM, iterations are enqueued in batches, one read-back per batch (default)
 *   HOST    fused passes, every scalar read back as it is formed          (env LIS_AMD_HOST_SCALARS=1)
 *   UNFUSED one kernel per reference call                                 (env LIS_AMD_NO_FUSION=1)  */
#define LIS_AMD_LOOP_DEVICE  0
#define LIS_AMD_LOOP_HOST    1
#define LIS_AMD_LOOP_UNFUSED 2
LIS_INT lis_amd_set_loop_mode(LIS_INT mode);
/* Reference-order reductions (parity mode; env LIS_AMD_REFERENCE_REDUCTIONS=T).  T > 0: lis_vector_dot / nrm2 / nrm1 / sum and every sum
 * a Krylov loop forms are added as the reference's OpenMP build adds them with OMP_NUM_THREADS = T -- T contiguous chunks by LIS_GET_ISIE, each
 * strictly left to right from 0.0, the T partial sums added serially (src/vector/lis_vector_ops.c:88-107, :241-259) -- so iteration counts, residual
 * histories and solutions carry the reference's bits (tests/test_reference_order_gpu.py demands equality with oracle/_ref).  The products' fused
 * dot epilogues are not taken while it is on.  A job of P ranks at T = 1 forms the sums of one rank at T = P (row blocks by LIS_GET_ISIE, fold in
 * rank order).  0 (default): the fixed trees, whose results are reproducible but not the reference's last bits.  Slow: one lane adds per chunk. */
LIS_INT lis_amd_set_reference_reductions(LIS_INT T);
LIS_INT lis_amd_get_reference_reductions(void);
/* 1 when the last lis_solve ran CG + Jacobi on a matrix with a constant diagonal and its fused passes took 1/diag as one double
 * instead of reading the array (bit-identical; LIS_AMD_NO_UNIFORM_JACOBI=1 switches it off), else 0 */
LIS_INT lis_amd_last_solve_uniform_jacobi(void);
/* hipGraph replay of the device-driven Krylov batches (opt-in, environment LIS_AMD_GRAPHS=1): from the second full batch on
 * a single-rank solve replays one captured batch instead of launching its kernels.  Results do not depend on it (same
 * kernels, arguments and order); measured it does not pay on this hardware -- a small system's iteration is bound by the
 * GPU's ~4.5 us per dependent kernel, not by the host's launch rate (profiles/r02_graph_sweep.txt). */
LIS_INT lis_amd_set_graphs(LIS_INT on);
LIS_INT lis_amd_last_solve_graph_replays(void);        /* batches of the last lis_solve that ran as a graph replay */
LIS_INT lis_amd_last_solve_renumbered(void);           /* 1: the last lis_solve ran in the numbering of a reordered plan (b, x0 gathered once, x scattered back; lis_amd_matrix_reordered) */

/* vectors */
LIS_INT lis_amd_vector_sync_host(LIS_VECTOR v);        /* make v->value[] current (D2H if needed)        */
LIS_INT lis_amd_vector_host_modified(LIS_VECTOR v);    /* v->value[] was written directly: HBM copy stale */
LIS_INT lis_amd_vector_device_ptr(LIS_VECTOR v, LIS_SCALAR **dptr);  /* current HBM copy (uploads if needed) */
LIS_INT lis_amd_vector_device_modified(LIS_VECTOR v);  /* HBM copy was written by foreign kernels          */

/* matrices */
LIS_INT lis_amd_matrix_upload(LIS_MATRIX A);           /* build the HBM copy now (otherwise on first use) */
LIS_INT lis_amd_matrix_host_modified(LIS_MATRIX A);   /* host arrays changed: drop the HBM copy */
/* Host writes to a matrix's arrays after its HBM copy was built.  The reference adopts arrays (src/matrix/lis_matrix_csr.c:98-103) and reads them live on every
 * product.  Here: arrays that came from lis_matrix_malloc_<fmt> (lis_matrix_csr.c:170), from element-wise assembly or from a conversion live on pages of the
 * library's own; under the default (lazy) coherence a write to them is SEEN -- one page fault -- and the copy, its plan and the transposed operator are rebuilt
 * before the next product, with no call from the program.  Arrays the caller malloc'ed and handed to lis_matrix_set_<fmt> cannot be watched: call
 * lis_amd_matrix_host_modified(A) after changing them, or run with LIS_AMD_MATRIX_CHECK=1 / lis_amd_set_matrix_check(1) while debugging: every use of a matrix
 * then re-hashes its host arrays (a pass over them on the host cores) and rebuilds a stale copy, with one line on stderr naming the matrix.
 * lis_amd_matrix_protected_arrays: how many arrays of A are write-protected right now (tests).  System calls that WRITE INTO a protected array (fread into
 * A->value) fail with EFAULT, as for vectors: call lis_amd_matrix_host_modified(A) first, which opens the pages. */
/* 0 (env LIS_AMD_PLAIN_MALLOC=1): lis_matrix_malloc_<fmt> returns plain malloc memory, as the reference's lis_malloc does -- for a program that free()s those arrays
 * itself or passes them to read(2) / fread / MPI_Recv between solves (a system call into a write-protected page fails with EFAULT).  Such arrays are never protected and
 * an in-place edit is not seen: lis_amd_matrix_host_modified(A) is then the contract, as for any array the caller malloc'ed.  1 (default): pages of the library's own,
 * released by lis_free / lis_matrix_destroy only, read-only while the HBM copy lives. */
LIS_INT lis_amd_set_matrix_pages(LIS_INT on);
LIS_INT lis_amd_set_matrix_check(LIS_INT on);
LIS_INT lis_amd_matrix_protected_arrays(LIS_MATRIX A);
LIS_INT lis_amd_matrix_host_written(LIS_MATRIX A);     /* 1: a host write to one of A's watched arrays was seen since the HBM copy was built (tests) */
LIS_INT lis_amd_matrix_page_test_watch(LIS_MATRIX A);  /* tests without a GPU: adopt + write-protect A's arrays as an upload would; returns the number protected */
/* number of entries of the column-offset dictionary when the HBM copy of A carries one-byte column codes (liship.h
 * "index coding": matrices on <= 255 diagonals), 0 when it reads the 4 B indices; uploads A if needed */
LIS_INT lis_amd_matrix_index_codes(LIS_MATRIX A);
/* number of row patterns when the HBM copy of A keeps ONE byte per row (liship.h "row patterns": the rows of a coded matrix
 * follow <= 255 offset sequences), 0 otherwise; uploads A if needed */
LIS_INT lis_amd_matrix_row_patterns(LIS_MATRIX A);
/* 1 when those patterns are also kept as 32 B records (1..7 offsets each: the 7-point stencil) and the products use the
 * kernel that issues its x gathers ahead of the value slice, 0 otherwise; uploads A if needed */
LIS_INT lis_amd_matrix_pattern_records(LIS_MATRIX A);
/* 1 (2: rows of 8..32 entries, the wide form) when the rows of each pattern also carry the same values (a constant-coefficient stencil) and the HBM copy keeps them in
 * the pattern records -- the products then read ONE byte per row of matrix data, neither values nor indices (liship.h "value
 * records"; LIS_AMD_NO_VALUE_RECORDS=1 switches them off) -- 0 otherwise; uploads A if needed.  Like every other part of the HBM
 * copy they follow the host arrays only through lis_amd_matrix_host_modified(); the arrays of a matrix adopted with
 * lis_amd_matrix_set_csr_device() must not be rewritten in place once a product has run. */
LIS_INT lis_amd_matrix_value_records(LIS_MATRIX A);
/* 1 when, on top of the value records, ONE pattern carries at least half of the rows and the others are read in its slots: the products
 * then issue that pattern's x gathers together with the pattern bytes -- one round trip per wavefront, no LDS, no barrier (liship.h
 * "dominant pattern") -- 0 otherwise; uploads A if needed */
LIS_INT lis_amd_matrix_dominant_pattern(LIS_MATRIX A);
/* 1 when the HBM copy's rows of 8..32 entries ride in wide value records AND the plan stages x per wavefront for a dominant pattern -- one pattern with half of
 * the rows, or the union of the patterns the rows take turns on (b x b blocked stencils kept row by row) -- 0 otherwise; uploads A if needed */
LIS_INT lis_amd_matrix_wide_dominant(LIS_MATRIX A);
/* b when the HBM copy of a b x b BSR matrix with constant coefficients is its row form AND the product gives a lane a whole block row (each x read once for the
 * block row's b sums, liship.h "block rows"), else 0; uploads A if needed */
LIS_INT lis_amd_matrix_block_rows(LIS_MATRIX A);
/* the form of the whole local product of a 7-point matrix with value records: 0 = the gathering dominant-pattern kernel, 1 = the z-marching kernel (each x loaded once per
 * plane tile; liship.h), 2 = its box form, which reads no pattern byte either (x and y alone are streamed: 16 B per row); uploads A if needed */
/* round 5: 3 = the 27-point box stencil with constant coefficients marches (spmv_csr_box27_march_kernel), 4 = the 7-point stencil in 2 x 2 blocks does
 * (a constant-coefficient BSR matrix in its row form: spmv_csr_block2_march_kernel) */
LIS_INT lis_amd_matrix_marching(LIS_MATRIX A);
/* rows per plane of the structured grid the plan found (the largest pattern offset, or the band of a matrix without row patterns): the XCD strips of the kernels that
 * stream the matrix are cut from it (liship.h); 0 = none, natural block order; uploads A if needed */
LIS_INT lis_amd_matrix_strip_rows(LIS_MATRIX A);
/* ELL and DIA matrices with constant coefficients are kept in HBM as CSR rows that list the format's terms in the format's order
 * (bit-identical sums), so that the value records apply.  0 keeps the native ELL / DIA layout and kernels for matrices uploaded
 * from now on (env LIS_AMD_NO_ROW_FORM=1): A/B measurements, and the tests that pin the native kernels at full size. */
LIS_INT lis_amd_set_row_form(LIS_INT on);
/* REFERENCE LAYOUT mode (env LIS_AMD_REFERENCE_LAYOUT=1), a supported mode: every product streams the reference's own arrays -- CSR: 4 B index[] + 8 B value[] per
 * non-zero + ptr[] (the loop of src/matvec/lis_matvec_csr.c:97-109 on 12 B per non-zero + 20 B per row, the count the CSR roofline target is quoted on), ELL / DIA / BSR
 * their native arrays -- and nothing a plan could derive from them (one-byte column codes, row patterns, value records, block-local columns, renumbering, row forms).
 * Results carry the same bits; bench.py's headline runs in this mode.  Takes effect for plans already built and for matrices uploaded from now on. */
LIS_INT lis_amd_set_reference_layout(LIS_INT on);
LIS_INT lis_amd_get_reference_layout(void);
/* storage type of the HBM copy (LIS_MATRIX_CSR for matrices re-laid as rows: CSC, JAD, the row form above), uploads A if needed */
LIS_INT lis_amd_matrix_device_type(LIS_MATRIX A);
/* total length of the per-row-block lists of distinct columns when the HBM copy of A carries block-local columns (liship.h:
 * long rows that share their columns), 0 when it does not; uploads A if needed */
LIS_INT lis_amd_matrix_local_columns(LIS_MATRIX A);
/* total length of those lists in the REORDERED form, when the plan renumbered rows and columns because the caller's numbering has no locality (liship.h:
 * liship_csr_plan_reorder; one rank, CSR with long rows; env LIS_AMD_NO_REORDER=1 keeps the caller's numbering); 0 when it did not; uploads A if needed */
long long lis_amd_matrix_reordered(LIS_MATRIX A);
/* WHEN the renumbered form is built: by the first lis_solve that finds A's HBM copy has served `products` products in the caller's numbering (default 4096; env
 * LIS_AMD_REORDER_AFTER).  Building it -- the numbering found on the device (breadth-first distances from landmarks, Morton keys, a radix sort: kernels/csr_order.hpp; rounds
 * 4-5 walked the graph on the host), P A P^T and its plan in HBM -- costs 0.17 s and +3.5 GB on the Queen-class matrix and saves 0.06-0.1 ms per iteration there: a program
 * earns it back after ~3000 iterations, and the first solves of most programs take 40-50.  A matrix whose block-local lists FAILED altogether (no locality at all: an
 * unstructured mesh numbered at random inside coarse cells runs at 30-40 % of its roofline and twice as fast renumbered) waits for products / 16 only (256): about what
 * building the form costs in products of that kind.  0: at plan time (upload / assemble), the round-5 behaviour. */
LIS_INT lis_amd_set_reorder_after(long long products);
long long lis_amd_matrix_products_served(LIS_MATRIX A);      /* products A's current HBM copy has served (approximately: a fused product + dot that falls back counts twice) */
/* the liship plan of A's HBM copy when it is served as CSR rows (for the liship_csr_plan_* queries of liship.h; owned by A), else NULL; uploads A if needed */
void *lis_amd_matrix_csr_plan(LIS_MATRIX A);
/* adopt CSR arrays that already live in HBM (no host copy exists; A must be sized and unassembled).
 * ptr has n+1 entries, columns are local (0..np-1, ghosts >= n).  The arrays are freed with the matrix. */
LIS_INT lis_amd_matrix_set_csr_device(LIS_INT nnz, LIS_INT np, LIS_INT *dptr, LIS_INT *dindex,
                                      LIS_SCALAR *dvalue, LIS_MATRIX A);
/* 3-D 7-pt Poisson on an l x m x n grid (test/test3.c:114-127; sorted!=0: test/spmvtest3.c:192-195),
 * generated in HBM for this rank's rows; A must be created + set_size'd with global size l*m*n.
 * In a multi-GPU job the halo tables are derived in closed form (whole planes per rank). */
LIS_INT lis_amd_matrix_poisson3d(LIS_MATRIX A, LIS_INT l, LIS_INT m, LIS_INT n, LIS_INT sorted);
/* b = A*1 for that matrix in closed form (test/test3.c:150) */
LIS_INT lis_amd_vector_poisson3d_rhs(LIS_VECTOR b, LIS_INT l, LIS_INT m, LIS_INT n);

/* seconds of the last lis_input: the whole call, and its lis_matrix_assemble step alone -- for a matrix that lives in HBM that step is the upload and the plan */
LIS_INT lis_amd_last_input_times(double *total_s, double *assemble_upload_plan_s);

/* stream all library work is queued on (hipStream_t as void*), and a full device sync */
void   *lis_amd_stream(void);
LIS_INT lis_amd_trim(void);                            /* give the solver work-vector pool back to the driver */
LIS_INT lis_amd_synchronize(void);

/* ---- multi-GPU: one process per GPU, row-block partition (LIS_GET_ISIE), RCCL over xGMI ------------
 * Replaces the MPI layer of the reference (src/matrix/lis_matrix_mpi.c): halo = packed export rows
 * exchanged with grouped ncclSend/ncclRecv straight into x[n..np), reductions = all-gather of the
 * per-rank partial sums folded in rank order (deterministic).
 *
 * Bootstrap: rank 0 calls lis_amd_comm_get_unique_id(), the launcher broadcasts the 128 bytes (any
 * channel: torch.distributed/gloo in bench.py, a file, MPI), every rank calls lis_amd_comm_init_rccl().
 * lis_amd_comm_init_callbacks() installs host-memory collectives instead (CPU tests with gloo). */
#define LIS_AMD_UNIQUE_ID_BYTES 128
LIS_INT lis_amd_comm_get_unique_id(void *id128);
LIS_INT lis_amd_comm_init_rccl(const void *id128, LIS_INT rank, LIS_INT nprocs, LIS_INT device);
typedef struct {
    /* recv[r*count .. (r+1)*count) <- send of rank r, for all r (host memory) */
    int (*allgather)(void *ctx, const void *send, void *recv, size_t bytes_per_rank);
    /* exchange with every neighbour: send sendbuf[sptr[i]..sptr[i+1]) doubles to rank neib[i], receive
     * recvbuf[rptr[i]..rptr[i+1]) from it (host memory) */
    int (*neighbor_exchange)(void *ctx, int nneib, const int *neib, const double *sendbuf, const int *sptr,
                             double *recvbuf, const int *rptr);
    void *ctx;
} lis_amd_comm_callbacks;
LIS_INT lis_amd_comm_init_callbacks(const lis_amd_comm_callbacks *cb, LIS_INT rank, LIS_INT nprocs);
/* the reference's lis_send_recv on a HOST array x[np] through the callback communicator (tests of the tables) */
LIS_INT lis_amd_halo_exchange_host(LIS_MATRIX A, LIS_SCALAR x[]);
LIS_INT lis_amd_comm_finalize(void);
LIS_INT lis_amd_comm_rank(void);
LIS_INT lis_amd_comm_size(void);
LIS_INT lis_amd_comm_kind(void);                       /* 0: no communicator, 1: RCCL, 2: host callbacks */
/* 1 when the halo exchange that overlaps the interior rows has an RCCL communicator of its own (formed inside lis_amd_comm_init_rccl; LIS_AMD_ONE_COMMUNICATOR=1
 * keeps everything on the first one, ordered by events).  LIS_AMD_COMM_TIMEOUT=<seconds> (default 300, 0 = off): a communicator that does not form, or a stream
 * that carries collectives and does not drain, within that time aborts the process with a message instead of hanging the job. */
LIS_INT lis_amd_comm_halo_communicator(void);
/* halo exchange overlapped with the interior rows of a product (default on; env LIS_AMD_NO_OVERLAP=1): for A/B runs */
LIS_INT lis_amd_set_overlap(LIS_INT on);
/* one halo exchange of x's ghost entries in HBM, by itself (what a product does before its boundary rows) */
LIS_INT lis_amd_halo_exchange(LIS_MATRIX A, LIS_VECTOR x);

#ifdef __cplusplus
}
#endif
#endif
