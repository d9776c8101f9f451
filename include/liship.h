/*
 * liship.h -- kernel-level C ABI of the MI355X (gfx950) hot path: plain device pointers and sizes.
 *
 * This is the layer a Lis maintainer binds from C: each entry point replaces one
 * OpenMP hot loop of the reference (anishida/lis 2.1.11; citations are file:line under /root/reference).
 * Everything here is hand-written HIP; there is no CPU fallback -- a call on a machine without a
 * gfx950 device fails with a HIP error code.
 *
 * Conventions
 *   - all array arguments are DEVICE pointers (hipMalloc'd), int32 indices, f64 scalars
 *     (LIS_INT / LIS_SCALAR of the reference's default build, include/lis.h:446,461);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - return value: 0 on success, otherwise the hipError_t of the failing runtime call, or
 *     LISHIP_ERR_ARG (-1) for an argument the kernel cannot serve;
 *   - arithmetic: one rounded multiply and one rounded add per term (no FMA contraction), rows summed
 *     strictly in stored order starting from +0.0 -- bit-identical to the reference's CPU loops.
 */
#ifndef LISHIP_H
#define LISHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISHIP_ERR_ARG (-1)

/* ------------------------------------------------------------------ device / memory utilities */
int  liship_device_count(int *count);
int  liship_set_device(int device);
int  liship_get_device(int *device);
int  liship_device_name(char *buf, int buflen);           /* e.g. "gfx950:sramecc+:xnack-" */
int  liship_malloc(void **dptr, size_t bytes);
int  liship_free(void *dptr);
int  liship_memset(void *dptr, int byte, size_t bytes, void *stream);
int  liship_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int  liship_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int  liship_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int  liship_stream_create(void **stream);
int  liship_stream_destroy(void *stream);
int  liship_stream_synchronize(void *stream);
/* watchdog for streams that carry collectives: with a limit > 0 liship_stream_synchronize polls and returns LISHIP_ERR_TIMEOUT (-2) when the stream has not
 * drained within `seconds` (a peer that never arrived); 0 = wait for ever (default) */
int  liship_set_sync_timeout(double seconds);
#define LISHIP_ERR_TIMEOUT (-2)
int  liship_device_synchronize(void);
/* hipGraph capture of whatever is enqueued on `stream` between begin and end (the Krylov loops replay a batch of
 * iterations this way when the system is small enough to be launch bound -- the reference has no counterpart: its
 * loops are host code, src/solver/lis_solver_cg.c:176-215).  capture_end always ends the capture; with exec == NULL the
 * recording is dropped.  A failed capture leaves the stream usable for plain launches. */
int  liship_graph_capture_begin(void *stream);
int  liship_graph_capture_end(void *stream, void **exec);
int  liship_graph_launch(void *exec, void *stream);
int  liship_graph_destroy(void *exec);
/* HIP-event stopwatch on `stream`: start/stop bracket a region, elapsed_ms reads it after stop+sync */
/* page-locked host memory for the scalar read-backs */
int  liship_malloc_host(void **ptr, size_t bytes);
int  liship_free_host(void *ptr);
/* stream-ordering events: record on one stream, make another stream wait (no host involvement) */
int  liship_event_create(void **event);
int  liship_event_destroy(void *event);
int  liship_event_record(void *event, void *stream);
int  liship_event_synchronize(void *event);
int  liship_stream_wait_event(void *stream, void *event);
/* the box's streaming yardstick: every workgroup sums `reads` (1, 2, 4, 8 or 13) consecutive 512-double tiles of src (reads * n doubles) into one tile of dst (n doubles),
 * nontemporal, no gather: what a kernel with the products' read : write ratio reaches on THIS box (bench.py reports it beside the product's roofline fraction); n % 512 == 0 */
int  liship_stream_yardstick(int reads, size_t n, const double *src, double *dst, int wgs_per_cu /* 0: a workgroup per tile, else persistent workgroups per CU */, void *stream);
int  liship_timer_create(void **timer);
int  liship_timer_destroy(void *timer);
int  liship_timer_start(void *timer, void *stream);
int  liship_timer_stop(void *timer, void *stream);
int  liship_timer_elapsed_ms(void *timer, float *ms);
const char *liship_error_string(int code);

/* ------------------------------------------------------------------ CSR SpMV
 * replaces lis_matvec_csr, src/matvec/lis_matvec_csr.c:90-110 (unsplit branch):
 *     y[i] = sum_{j=ptr[i]}^{ptr[i+1]-1} value[j] * x[index[j]]      (in stored order)
 *
 * A plan is the nnz-balanced row split of one matrix (merge-path coordinates row+nnz, whole rows per
 * workgroup); it depends only on ptr[] and is built once per matrix on the device. */
typedef struct liship_csr_plan_s *liship_csr_plan_t;
int  liship_csr_plan_create(liship_csr_plan_t *plan, int n, const int *ptr, void *stream);
/* Index coding (setup-time, optional, never an error when the matrix does not qualify): if the entries of the matrix sit
 * on at most 255 distinct diagonals -- every structured-grid discretisation -- the plan keeps ONE byte per non-zero
 * (the position of column - row in a sorted dictionary) and the row-gather kernel streams 9 B per non-zero instead
 * of 12.  Same terms in the same order: results are bit-identical.  The index array must stay valid (long rows, array
 * tails and row-range launches may still read it).  liship_csr_plan_coded: dictionary size, 0 if not coded.
 * liship_spmv_csr_set_index_codes(0) makes every product ignore the codes (A/B measurements). */
int  liship_csr_plan_encode_indices(liship_csr_plan_t plan, const int *ptr, const int *index, void *stream);
int  liship_csr_plan_coded(liship_csr_plan_t plan);
/* Row patterns (setup-time, optional, after liship_csr_plan_encode_indices; never an error when the matrix does not qualify):
 * when the rows of a coded matrix follow at most 255 distinct (length, offset sequence) patterns -- a 7-point stencil has 27 --
 * the plan keeps ONE byte per row (its pattern) and a 2 B row start relative to the row block instead of one byte per
 * non-zero and the 4 B row pointer: 75 instead of 83 B per stencil row.  Same terms in the same order: bit-identical.
 * liship_csr_plan_row_patterns: number of patterns, 0 if none.  liship_spmv_csr_set_row_patterns(0): A/B switch.
 * When every pattern has 1..7 offsets (and there are at most 64, n < 2^29) the plan also keeps them as 32 B records and the
 * products run through the kernel that issues its x gathers ahead of the value slice: liship_csr_plan_pattern_records = 1. */
int  liship_csr_plan_encode_row_patterns(liship_csr_plan_t plan, const int *ptr, void *stream);
int  liship_csr_plan_row_patterns(liship_csr_plan_t plan);
int  liship_csr_plan_pattern_records(liship_csr_plan_t plan);
/* When the longest pattern has 8..32 offsets (the 27-point stencil) the plan keeps 144 B records (32 byte offsets, length) and the
 * values-streamed product runs four lanes per row, the gathers leaving ahead of the value slice and the row's one ordered sum handed
 * from lane to lane in registers (spmv_csr_pattern_team_kernel): liship_csr_plan_team_records = 1. */
int  liship_csr_plan_team_records(liship_csr_plan_t plan);
/* ... and when ONE pattern carries most rows and its sorted offsets are runs of consecutive columns of one length (the box stencils), the x a
 * wavefront's 16 neighbouring rows need is staged in LDS by a few coalesced loads instead of a gather per entry (spmv_csr_pattern_team_staged_kernel):
 * liship_csr_plan_team_form = 2 (1: the gathers of the first form; 0: no team records). */
int  liship_csr_plan_team_form(liship_csr_plan_t plan);
/* 1 when a plan with WIDE value records (constant-coefficient rows of up to 32 entries) also found a dominant pattern and runs the kernel with x staged per
 * wavefront and that pattern's slots and values in scalar registers (spmv_csr_valuerecw_staged_kernel) */
int  liship_csr_plan_wide_dominant(liship_csr_plan_t plan);
/* 1 when, on top of that, the matrix is the 27-point box stencil with constant coefficients on a grid that IS a box (lines of any even length from 128 on -- round 5: partial tiles --, rows in ascending
 * column order; checked row by row at plan time) and its whole-matrix product walks the planes of 128-column tiles with each x loaded once (round 5:
 * spmv_csr_box27_march_kernel -- x and y alone are streamed); liship_spmv_csr_set_dom_march(0) keeps the staged kernel (A/B), 2 marches at any size (tests) */
int  liship_csr_plan_box27(liship_csr_plan_t plan);
/* 0 when the products of this plan run a kernel with a row split of its own (the team / staged kernels) under the switches in force: the fused entry points
 * (liship_spmv_csr_dot_f64, liship_spmv_csr_rows_dot_f64) then refuse with LISHIP_ERR_ARG and the caller runs the product and one reduction pass */
int  liship_csr_plan_fused_dots(liship_csr_plan_t plan);
long long liship_csr_plan_fused_slots(liship_csr_plan_t plan);   /* upper bound of the reduction slots the fused product needs in up to three row ranges */
int  liship_spmv_csr_set_team(int on);           /* A/B switch: 0 = the one-lane-per-row pattern kernel for these rows too (same bits) */
/* Block rows on top of the wide value records: the rows b i .. b i + b - 1 of the matrix list the SAME columns in the same order (the row form of a b x b BSR matrix,
 * liship_bsr_to_rows; b = 2, 3, 4).  When one block row pattern carries half of the block rows, the product gives a LANE a block row: each x read once from the
 * wavefront's staged window feeds the b running sums, the b x len values are kernel arguments (spmv_csr_blockrows_staged_kernel).  Optional, never an error when the
 * matrix does not qualify; row ranges that cut a block row run the row-by-row kernels.  Bits of lis_matvec_bsr.c:293-343. */
int  liship_csr_plan_encode_block_rows(liship_csr_plan_t plan, int b, const int *ptr, void *stream);
int  liship_csr_plan_block_rows(liship_csr_plan_t plan);       /* b when the plan keeps them, else 0 */
/* 1 when those block rows are the 7-point stencil in 2 x 2 blocks on a grid that is a box (lines of any even length from 128 on -- round 5: partial tiles --; checked block row by block row at plan time)
 * and the whole-matrix product walks the planes with each x loaded once (round 5: spmv_csr_block2_march_kernel); liship_spmv_csr_set_dom_march as for the others */
int  liship_csr_plan_block2_march(liship_csr_plan_t plan);
int  liship_spmv_csr_set_block_rows(int on);                   /* A/B switch: 0 = the row-by-row kernels (same bits) */
/* The z-marching form of the dominant-pattern product (7-point stencil with value records, grid lines of any even length from 128 on: the last 128-column tile of a line may be partial): 1 = on for launches of 64 workgroups and more (default), 0 = the gathering kernel, 2 = on at any size,
 * 3 = at any size, but with the faces' masks even where plan time found the grid a box (liship_csr_plan_box_planes: planes in which a slot is missing exactly where its neighbour
 * lies outside the grid -- there the kernel reads no pattern byte at all).  Same bits in every form. */
int  liship_spmv_csr_set_dom_march(int on);
int  liship_csr_plan_box_planes(liship_csr_plan_t plan);
int  liship_csr_plan_marching(liship_csr_plan_t plan);       /* the whole-matrix product: 0 = the gathering kernels, 1 = z-marching with the faces' masks, 2 = its box form (no pattern bytes) */
int  liship_spmv_csr_set_wide_union(int on);     /* plan-time A/B switch: 0 = no virtual dominant pattern (a common supersequence of the patterns rows take turns on: b x b blocked stencils), 1 = from 2^19 rows on (default), 2 = at any size */
/* Value records (setup-time, optional, after liship_csr_plan_encode_row_patterns; never an error when the matrix does not
 * qualify): when the plan has 32 B pattern records and every row of a pattern carries the same values bit for bit -- a
 * constant-coefficient stencil -- the 7 values join the 7 offsets in the record and the products read neither the value nor
 * the index array: one byte per row (the idea of CSR-VI, value indexing, applied to whole rows).  Same products in the same
 * order: bit-identical.  The value array must not change afterwards (a new plan is needed if it does).
 * Rows of one offset pattern that carry different values (a Dirichlet row stored with the interior row's sparsity) split the pattern.
 * Patterns of 8..32 offsets (no 32 B records: the 9-point stencil in 2-D, the 19- and 27-point ones in 3-D) get WIDE records when
 * there are at most 48 of them after the same splitting (400 B per pattern).
 * liship_csr_plan_value_records: 1 if the plan has them, 2 for the wide form.  liship_spmv_csr_set_row_values(0): A/B switch
 * (stream the values). */
int  liship_csr_plan_encode_row_values(liship_csr_plan_t plan, const int *ptr, const double *value, void *stream);
int  liship_csr_plan_value_records(liship_csr_plan_t plan);
/* Dominant pattern (found with the value records, nothing to call): when ONE pattern carries at least half of the rows -- the interior
 * row of a stencil: 98.8 % of the rows at 512^3 -- the plan names it, and keeps for every other pattern which of ITS slots that pattern
 * has and the pattern's values there (a boundary row is the interior row minus the neighbours that do not exist: a subsequence).  The
 * products then take the dominant pattern's offsets and values as kernel ARGUMENTS and issue its x gathers together with the load of
 * the pattern bytes -- one round trip per wavefront, no records in LDS, no barrier; wavefronts that meet another pattern fetch its 64 B
 * slot record by scalar loads, rows whose pattern is no subsequence (ghost columns of a partitioned matrix) or whose speculative
 * addresses would leave x take their own records row by row.  Same products in the same order: bit-identical.
 * liship_csr_plan_dominant_pattern: 1 if the plan has one. */
int  liship_csr_plan_dominant_pattern(liship_csr_plan_t plan);
int  liship_spmv_csr_set_row_values(int on);
int  liship_spmv_csr_set_row_patterns(int on);
int  liship_spmv_csr_set_index_codes(int on);
/* Block-local columns (setup-time, optional, never an error when the matrix does not qualify): for matrices with long rows
 * (the plan's products kernel) whose row blocks address few distinct columns -- several unknowns per node, wide bands --
 * the plan keeps per row block the sorted list of its distinct columns and per non-zero a 2 B position in that list; the
 * kernel then gathers x once per distinct column into LDS and forms the products from LDS (10 B + 4 B x distinct/entries
 * per non-zero instead of 12, a fraction of the gathers).  Same terms in the same order: bit-identical.  The index array
 * must stay valid.  liship_csr_plan_localized: total length of the lists, 0 if the plan has none.
 * liship_spmv_csr_set_local_columns(0) makes every product ignore them (A/B measurements). */
int  liship_csr_plan_localize_columns(liship_csr_plan_t plan, const int *ptr, const int *index, void *stream);
long long liship_csr_plan_localized(liship_csr_plan_t plan);
int  liship_spmv_csr_set_local_columns(int on);
/* round 5: when every list is made of TRIPLES of consecutive columns (meshes with 3 unknowns per node) the plan also keeps the triples' first columns and the kernel
 * reads one 4 B run start per triple instead of three columns (a third of the list bytes).  liship_csr_plan_local_runs: 3 when the plan has them, else 0;
 * liship_spmv_csr_set_local_runs(0): A/B switch (the full lists), same bits */
int  liship_csr_plan_local_runs(liship_csr_plan_t plan);
int  liship_spmv_csr_set_local_runs(int on);
/* 0: the block-local kernel keeps one entry per lane and step (eight 2 B position loads) instead of pairs of neighbouring entries (four 4 B loads, 16 B LDS accesses): A/B, same bits */
int  liship_spmv_csr_set_local_pairs(int on);
/* 0: plans of short rows (mean < 22 entries, no column codes) never try block-local columns (the rule of rounds 2-5); default 1 (round 6): they are tried from a mean
 * of 4 entries per row on and kept under the long rows' rule -- unstructured meshes with one unknown per node: +6 .. +39 % over the lane-per-row kernel.  Same bits. */
int  liship_spmv_csr_set_local_short_rows(int on);
/* Reordering (round 5): when the lists of a plan with block-local columns are long -- more than one listed column per `min_items_per_listed` non-zeros (0: the
 * default, 4) -- or a long-row plan could not have lists at all (more than 2048 distinct columns per row block): the signs of a numbering without locality -- the plan
 * renumbers rows and columns by graph distances found on the device (breadth-first searches from 3-4 landmarks, Morton keys, a stable radix sort: csr_order.hpp; rounds 4-5: a Cuthill-McKee walk on the host), builds P A P^T in HBM (the same entries
 * in the same in-row order) with a plan of its own, and keeps it when that plan lists at most 3/4 of the columns.  Short rows (plans of the row-gather kernel, no lists):
 * the 128 B lines of x a row block touches are counted instead -- more than one per 4 entries starts the walk, at most half of them afterwards keeps its result.
 * WHO USES IT: liship_csr_plan_reordered_form hands P A P^T out as a matrix of its own, for callers that keep whole iterations in the new numbering (lis_solve).
 * Products through the original plan keep the caller's numbering by default: a single product would have to gather x into the numbering and scatter y out of it, one
 * random access per node each, which costs what the better kernel gains (Queen class) or several times the product (scalar unknowns, short rows).
 * liship_spmv_csr_set_reorder(2) lets whole-matrix products of LONG-row plans take it all the same (row r stored where the original row lives: every y[i] is the
 * reference's sum, lis_matvec_csr.c:97-109, term by term -- the same bits; row-range products and the fused reductions keep the original numbering, and
 * liship_csr_plan_fused_dots returns 0); 1 is the default, 0 switches the reordered form off altogether (A/B).  The values are copied: a matrix whose value[] changes
 * needs a new plan, as with value records.  Never an error when the matrix does not qualify (fewer than 65 536 rows, coded indices, short lists, columns outside
 * [0, n)); 2 = out of memory, the plan unchanged.  liship_csr_plan_reordered: listed columns (lines of x for short rows) of the reordered form, 0: none. */
int  liship_csr_plan_reorder(liship_csr_plan_t plan, const int *ptr, const int *index, const double *value, int min_items_per_listed, void *stream);
/* the same with a permutation to try first (HOST, n entries: new position -> row), e.g. the one a plan of the same sparsity pattern found: a matrix whose values were
 * edited needs a new plan but not a new walk.  A hint that is not a permutation, or does not shorten the lists enough, is dropped for a walk.
 * liship_csr_plan_reorder_permutation: the permutation of the reordered form to the host (LISHIP_ERR_ARG when the plan has none). */
int  liship_csr_plan_reorder_with(liship_csr_plan_t plan, const int *ptr, const int *index, const double *value, int min_items_per_listed, const int *perm_hint, void *stream);
int  liship_csr_plan_reorder_permutation(liship_csr_plan_t plan, int *perm_host);
/* A rank's local matrix of a multi-rank job (columns [n, ncols) are ghost columns, lis_matrix_mpi.c:274-306): said before liship_csr_plan_reorder, the numbering is found on
 * the owned columns alone, ghost columns keep their numbers in P A P^T and the rows that read one are placed behind all the others --
 * rows [0, liship_csr_plan_reordered_inner_rows) of the reordered form touch no ghost column (they can run while the halo travels). */
int  liship_csr_plan_set_ghost_columns(liship_csr_plan_t plan, int ncols);
int  liship_csr_plan_reordered_inner_rows(liship_csr_plan_t plan);
/* 1: the plan wanted block-local column lists and has none (lists too long / trial failed): a numbering without any locality */
int  liship_csr_plan_lists_failed(liship_csr_plan_t plan);
/* out[k] = the position of row index[k] under the permutation perm (perm[new position] = row; n entries): a list of rows -- a halo export list -- in the new numbering */
int  liship_permute_rows_of_list(int n, const int *perm, int count, const int *index, int *out, void *stream);
long long liship_csr_plan_reordered(liship_csr_plan_t plan);
int  liship_spmv_csr_set_reorder(int mode);
/* the reordered form as a matrix of its own -- P A P^T: its plan (owned by `plan`), arrays and the permutation perm[new position] = original row -- for callers that keep
 * whole iterations in the new numbering (lis_solve gathers b and x0 once and scatters x back at the end: no per-product passes); LISHIP_ERR_ARG when the plan has none
 * (or it is switched off).  liship_permute_gather_f64: xp[i] = x[perm[i]]; liship_permute_scatter_f64: x[perm[i]] = xp[i] (x and xp distinct). */
int  liship_csr_plan_reordered_form(liship_csr_plan_t plan, liship_csr_plan_t *inner, const int **ptr, const int **index, const double **value, const int **perm);
int  liship_permute_gather_f64(int n, const int *perm, const double *x, double *xp, void *stream);
int  liship_permute_scatter_f64(int n, const int *perm, const double *xp, double *x, void *stream);
/* round 4: the block-local kernel keeps the 2 B positions in registers and stages 3584 items (lists <= 1024 columns) or 3072 items (longer lists) per
 * workgroup -- 39.5 KB of LDS, four workgroups per CU.  0: plans built from now on take the round-3 form (4096-item blocks, positions through LDS: three /
 * two workgroups per CU); A/B measurements, same bits either way */
int  liship_spmv_csr_set_local_register_positions(int on);
/* structured grids, values streamed (spmv_csr_pattern7_kernel): every XCD takes one eighth of every plane of the grid and walks the planes in order, so that the
 * +-plane neighbours of its rows stay in its own L2 (round 4).  0: the natural block order (A/B measurements); same bits either way */
/* the same strips for the kernels that stream index[] (spmv_csr_rowgather_kernel: the reference's 12 B per non-zero; spmv_csr_coded_kernel): a plan
 * without row patterns learns its plane from the band of the matrix -- the largest |column - row|, when at least half of the rows reach it (two passes over
 * index[] at plan time; optional, never an error when the matrix does not qualify).  liship_csr_plan_strip_rows: rows per plane, 0 = natural order */
int  liship_csr_plan_scan_band(liship_csr_plan_t plan, const int *ptr, const int *index, void *stream);
int  liship_csr_plan_strip_rows(liship_csr_plan_t plan);
int  liship_spmv_csr_set_xcd_strips(int on);
/* On by default (round 6), NOT bit-identical to the reference for the rows it touches: the part of a row beyond the LDS stage (~2100 entries) is added
 * by a workgroup-wide tree per pass instead of one left-to-right chain (a 200 000-entry row is otherwise a 200 000-long
 * dependent add chain).  Deterministic, within 1e-14 of the row's magnitude; rows that fit the stage keep the reference's bits.
 * 0 = the chain: the reference's bits for every row (LIS_AMD_LONG_ROW_CHAIN=1; implied by the reference-order reductions mode). */
int  liship_spmv_csr_set_long_row_tree(int on);
int  liship_spmv_csr_set_row_block_dots(int on);  /* 1: fused dots of the dominant-pattern product as the row blocks' partial sums (the other forms' bits); LIS_AMD_ROW_BLOCK_DOTS=1 */
int  liship_spmv_csr_switches(void);            /* bit 0 team kernels on, bit 1 row-block dots, bit 2 long-row tree (tests of the LIS_AMD_* variables) */
int  liship_spmv_csr_set_uniform_rows(int on);    /* A/B: 0 keeps the row sums of uniform-length wavefronts on the skewed schedule (same bits) */
int  liship_csr_plan_destroy(liship_csr_plan_t plan);
int  liship_csr_plan_info(liship_csr_plan_t plan, int *n, long long *nnz, int *nblocks);
/* Rows that START with their first product instead of being added to 0.0: the reference's split products compute
 * t0 = D[i]*x[i]; t0 += (L row); t0 += (U row)  (lis_matvec_csr.c:64-89 and the is_splited branches of the other formats).
 * With the flag on, every product of this plan starts its row sums at -0.0, which reproduces that sequence bit for bit
 * (signed zeros included) on rows laid out as [D, L entries, U entries]. */
int  liship_csr_plan_set_first_term_initialises(liship_csr_plan_t plan, int on);
int  liship_spmv_csr_f64(liship_csr_plan_t plan, const int *ptr, const int *index,
                         const double *value, const double *x, double *y, void *stream);
/* the product with a fused reduction epilogue: result[0] = sum_r w[r]*y[r] (w may be x: CG's <p,Ap>,
 * lis_solver_cg.c:185-188), result[1] = sum_r y[r]^2 when want_sumsq (BiCGSTAB's <t,s>,<t,t>,
 * lis_solver_bicgstab.c:262-268).  y is bit-identical to liship_spmv_csr_f64; saves a 16 B/row pass.
 * Returns LISHIP_ERR_ARG when it cannot serve the call (then use the plain product + liship_dot_f64). */
int  liship_spmv_csr_dot_f64(liship_csr_plan_t plan, const int *ptr, const int *index, const double *value,
                             const double *x, double *y, const double *w, int want_sumsq,
                             double *result, void *work, void *stream);
/* same product restricted to rows [row_begin,row_end) (used to overlap the halo exchange) */
int  liship_spmv_csr_rows_f64(liship_csr_plan_t plan, int row_begin, int row_end, const int *ptr,
                              const int *index, const double *value, const double *x, double *y,
                              void *stream);
/* The fused-reduction product of liship_spmv_csr_dot_f64 in parts: rows [row_begin,row_end) now, their per-row-block
 * partial sums parked in `work` from slot_base on (*slots_used of them); after the last part
 * liship_spmv_csr_dot_finish_f64 folds all parked partials into result[0..1].  A multi-rank job uses it to run the rows
 * that reference no ghost column while the halo travels (LIS_MATVEC_SENDRECV, include/lis_matvec.h:31-44, serialises
 * the two in the reference). */
int  liship_spmv_csr_rows_dot_f64(liship_csr_plan_t plan, int row_begin, int row_end, const int *ptr, const int *index,
                                  const double *value, const double *x, double *y, const double *w, int want_sumsq,
                                  void *work, int slot_base, int *slots_used, void *stream);
int  liship_spmv_csr_dot_finish_f64(int slots_used, int want_sumsq, double *result, void *work, void *stream);
/* tuning knobs for experiments (bench/profiling only): variant 0 = default */
int  liship_spmv_csr_set_variant(int variant);

/* ELL / DIA products with the reduction epilogue of liship_spmv_csr_dot_f64 (same contract and fallback rule) */
int  liship_spmv_ell_dot_f64(int n, int maxnzr, const int *index, const double *value, const double *x, double *y,
                             const double *w, int want_sumsq, double *result, void *work, void *stream);
/* ELL index coding (as liship_csr_plan_encode_indices): *codes / *dict are allocated on the device when the n*maxnzr
 * slots (padding included) sit on <= 255 diagonals and n is even, else stay NULL; free them with liship_free.
 * liship_spmv_ell_coded_f64: the ELL product from value + codes (9 B per slot), want_sumsq < 0 for the plain product,
 * 0 / 1 for the reduction epilogue of liship_spmv_ell_dot_f64; bit-identical to the 4 B-index kernels. */
int  liship_ell_encode_indices(int n, int maxnzr, const int *index, unsigned char **codes, int **dict, int *ndict, void *stream);
int  liship_spmv_ell_coded_f64(int n, int maxnzr, const unsigned char *codes, const int *dict, const double *value,
                               const double *x, double *y, const double *w, int want_sumsq, double *result,
                               void *work, void *stream);
int  liship_spmv_dia_dot_f64(int n, int ncols, int nnd, const int *offsets, const double *value, const double *x, double *y,
                             const double *w, int want_sumsq, double *result, void *work, void *stream);
/* ------------------------------------------------------------------ other formats
 * ELL   lis_matvec_ell  src/matvec/lis_matvec_ell.c:113-128   value/index column-major [maxnzr][n]
 * DIA   lis_matvec_dia  src/matvec/lis_matvec_dia.c:148-172   ONE-chunk layout value[d*n+i], offsets index[nnd]
 * JAD   lis_matvec_jad  src/matvec/lis_matvec_jad.c:170-196   ONE-chunk layout, y[perm[i]] = w[i]
 * BSR   lis_matvec_bsr* src/matvec/lis_matvec_bsr.c:57-858    column-major bnr x bnc blocks; like the
 *       reference it writes all nr*bnr rows, so y needs nr*bnr entries and x nc*bnc (the `pad`)
 * DIA   ncols = number of readable x entries (n, or np with ghost columns: lis_matvec_dia.c:158-162)
 * (CSC  lis_matvec_csc  src/matvec/lis_matvec_csc.c:128-144 is served by the CSR kernel on the
 *  column-ordered transpose the host layer builds at upload) */
int  liship_spmv_ell_f64(int n, int maxnzr, const int *index, const double *value,
                         const double *x, double *y, void *stream);
int  liship_spmv_dia_f64(int n, int ncols, int nnd, const int *offsets, const double *value,
                         const double *x, double *y, void *stream);
/* rows [row_begin, row_end) of the ELL / DIA product only (layouts of all n rows; codes / dict may be NULL): the parts a multi-rank
 * job launches around its halo exchange -- same bits as the whole launch */
int  liship_spmv_ell_rows_f64(int n, int maxnzr, const int *index, const unsigned char *codes, const int *dict, const double *value,
                              const double *x, double *y, int row_begin, int row_end, void *stream);
int  liship_spmv_dia_rows_f64(int n, int ncols, int nnd, const int *offsets, const double *value,
                              const double *x, double *y, int row_begin, int row_end, void *stream);
int  liship_spmv_jad_f64(int n, int maxnzr, const int *perm, const int *ptr, const int *index,
                         const double *value, const double *x, double *y, void *stream);
int  liship_spmv_bsr_f64(int nr, int bnr, int bnc, const int *bptr, const int *bindex,
                         const double *value, const double *x, double *y, void *stream);
/* the same with the number of stored blocks given (bnnz < 0: unknown), which lets the library choose between a lane
 * per block row (short block rows) and a lane per block (long ones) */
int  liship_spmv_bsr_nnz_f64(int nr, int bnnz, int bnr, int bnc, const int *bptr, const int *bindex,
                             const double *value, const double *x, double *y, void *stream);
/* block rows [brb, bre) only (the array layout is that of all nr block rows): the parts of a multi-rank product around its halo exchange -- same bits */
int  liship_spmv_bsr_rows_f64(int nr, int bnnz, int bnr, int bnc, const int *bptr, const int *bindex, const double *value,
                              const double *x, double *y, int brb, int bre, void *stream);
/* A/B switch: 0 keeps square blocks with long block rows (mean > 12 / 16 / 12 stored blocks per block row at 2x2 / 3x3 / 4x4) on the two-phase tile
 * kernels instead of the team-per-block-row kernel (spmv_bsr_team_kernel); same bits either way. */
int  liship_spmv_bsr_set_team(int on);
/* BSR product with the reduction epilogue (the contract of liship_spmv_csr_dot_f64): square blocks 2..4 and short block
 * rows; n = number of scalar rows (rows of the last block row beyond it are padding and stay out of the sums) */
int  liship_spmv_bsr_dot_f64(int nr, int n, int bnnz, int bs, const int *bptr, const int *bindex, const double *value,
                             const double *x, double *y, const double *w, int want_sumsq, double *result,
                             void *work, void *stream);

/* ------------------------------------------------------------------ vector kernels
 * element-wise: src/vector/lis_vector_opv.c (axpy :174, xpay :214, axpyz :253, scale :285, pmul :325,
 * pdiv :365, set_all :397, abs :428, reciprocal :458, shift :520); copy is a device memcpy. */
int  liship_axpy_f64 (int n, double alpha, const double *x, double *y, void *stream);       /* y += a*x    */
int  liship_xpay_f64 (int n, const double *x, double alpha, double *y, void *stream);       /* y = x + a*y */
int  liship_axpyz_f64(int n, double alpha, const double *x, const double *y, double *z, void *stream);
int  liship_scale_f64(int n, double alpha, double *x, void *stream);
int  liship_scale_to_f64(int n, double alpha, const double *x, double *y, void *stream);    /* y = a*x (lis_solver_gmres.c:290-296) */
int  liship_pmul_f64 (int n, const double *x, const double *y, double *z, void *stream);
int  liship_pdiv_f64 (int n, const double *x, const double *y, double *z, void *stream);
int  liship_set_all_f64(int n, double alpha, double *x, void *stream);
int  liship_abs_f64  (int n, double *x, void *stream);
int  liship_reciprocal_f64(int n, double *x, void *stream);
int  liship_shift_f64(int n, double sigma, double *x, void *stream);
int  liship_rsqrt_abs_f64(int n, double *x, void *stream);                                   /* x = 1/sqrt(|x|)   (lis_matrix_ops.c:611-614) */
/* rows of a CSR matrix scaled in HBM: value *= d[row] (symm 0, lis_matrix_csr.c:609-647) or value = value*d[row]*d[col] (symm 1, :651-690) */
int  liship_csr_scale_f64(int n, const int *ptr, const int *index, double *value, const double *d, int symm, void *stream);
/* fused forms: the SAME per-element expressions in the SAME order as the calls they replace (bit-identical),
 * one pass over HBM instead of two or three */
int  liship_axpy2_f64(int n, double a, const double *x, double b, const double *w, double *y, void *stream);       /* y += a*x; y += b*w   (lis_solver_bicgstab.c:272-273) */
int  liship_axpy_xpay_f64(int n, double a, const double *x, const double *w, double b, double *y, void *stream);   /* y += a*x; y = w + b*y (lis_solver_bicgstab.c:212-213) */
/* two-stage reductions (wavefront shuffle + LDS, then one fixed-order pass over the partials):
 * src/vector/lis_vector_ops.c dot :58-127, nrm2 :210-271, nrm1 :278-342, sum :418-478.
 * `result` is a DEVICE pointer to `count` doubles, `work` a device scratch of liship_reduce_work_bytes().
 * *_partial variants leave the un-rooted local sum (for a cross-GPU all-reduce before sqrt). */
size_t liship_reduce_work_bytes(void);
/* T > 0: the reductions below (and the fused update passes) add their terms in the reference's order for OMP_NUM_THREADS = T -- chunk t of T is
 * LIS_GET_ISIE(t, T, n) (include/lis.h:1067-1078), summed left to right from 0.0 by one lane, the T partials added serially in chunk order
 * (src/vector/lis_vector_ops.c:88-107) -- instead of the fixed tree; the products' fused-dot entry points then return LISHIP_ERR_ARG.  0: trees. */
int  liship_set_reference_reductions(int T);
int  liship_get_reference_reductions(void);
int  liship_dot_f64 (int n, const double *x, const double *y, double *result, void *work, void *stream);
int  liship_nrm2_f64(int n, const double *x, double *result, void *work, void *stream);
int  liship_sumsq_f64(int n, const double *x, double *result, void *work, void *stream);
int  liship_nrm1_f64(int n, const double *x, double *result, void *work, void *stream);
int  liship_sum_f64 (int n, const double *x, double *result, void *work, void *stream);
/* x += alpha*p; r += (-alpha)*q; result[0] = sum r^2   (lis_solver_cg.c:199-205: two axpys + the residual norm) */
int  liship_cg_update_f64(int n, double alpha, const double *p, const double *q, double *x, double *r,
                          double *result, void *work, void *stream);
/* the same with the Jacobi solve of the next iteration folded in: z = r.*dinv is formed per element (not
 * stored) and result = {sum r^2, sum r*z}   (lis_solver_cg.c:199-205 + :176-180 of the next iteration) */
int  liship_cg_update_jacobi_f64(int n, double alpha, const double *p, const double *q, const double *dinv,
                                 double *x, double *r, double *result, void *work, void *stream);
/* z = x.*d ; y = z + a*y   (Jacobi solve lis_precon_jacobi.c:121-124 + lis_vector_xpay, lis_solver_cg.c:183) */
int  liship_pmul_xpay_f64(int n, const double *x, const double *d, double a, double *y, void *stream);
/* y += a*x; result[0] = sum y^2            (lis_solver_bicgstab.c:233-236) */
int  liship_axpy_sumsq_f64(int n, double a, const double *x, double *y, double *result, void *work, void *stream);
/* y += a*x; result = {sum y^2, sum v*y}    (lis_solver_bicgstab.c:276-279 + :190 of the next iteration) */
int  liship_axpy_sumsq_dot_f64(int n, double a, const double *x, double *y, const double *v, double *result,
                               void *work, void *stream);
/* one modified Gram-Schmidt step of GMRES (lis_solver_gmres.c:219-227) with the coefficient read from HBM:
 * w += (-*hprev)*vprev, then result[0] = <w,vnext>, or sum w^2 when vnext is NULL.  hprev is a previous
 * result[], so a Hessenberg column needs no host synchronisation between its steps. */
int  liship_mgs_step_f64(int n, const double *hprev, const double *vprev, double *w, const double *vnext,
                         double *result, void *work, void *stream);
/* x *= 1/sqrt(*sumsq) with sumsq in HBM  (lis_vector_nrm2 + lis_vector_scale, lis_solver_gmres.c:229-232) */
int  liship_scale_inv_norm_f64(int n, const double *sumsq, double *x, void *stream);

/* ------------------------------------------------------------------ device-driven Krylov loops
 * The scalars of CG / BiCGSTAB (rho, alpha, beta, omega, the residual norm, the iteration count and the
 * convergence flag) can live in HBM: a block of LISHIP_KS_LEN doubles the host fills once.  The `_dev` forms of
 * the vector kernels read their coefficient(s) from that block at kernel start, reductions leave their sums in
 * it, and liship_krylov_step() -- one lane -- applies the reference's scalar statements between them
 * (lis_solver_cg.c:176-215, lis_solver_bicgstab.c:186-290, lis_solver_bicg.c:176-262) with the same IEEE operations the host loop would
 * use.  While a guard flag is installed (liship_krylov_guard), every vector / reduction / fused-product kernel
 * launched returns immediately if *flag != 0, so the host may enqueue a batch of iterations, read the block
 * back once, and find x, r and the iteration count exactly as the one-synchronisation-per-scalar loop leaves
 * them. */
enum {
	LISHIP_KS_RHO = 0, LISHIP_KS_RHO_OLD, LISHIP_KS_ALPHA, LISHIP_KS_NALPHA, LISHIP_KS_BETA, LISHIP_KS_OMEGA,
	LISHIP_KS_NOMEGA, LISHIP_KS_DOT0, LISHIP_KS_DOT1, LISHIP_KS_SUM0, LISHIP_KS_SUM1, LISHIP_KS_NRM2,
	LISHIP_KS_BNRM,          /* factor of the convergence test: 1/||b|| or 1/||r0|| (lis_solver.c:1792-1812) */
	LISHIP_KS_TOL,
	LISHIP_KS_ITER,          /* iterations completed (a double holding an integer) */
	LISHIP_KS_DONE,          /* 0 while iterating; the guard flag */
	LISHIP_KS_STATUS,        /* 0 running, 1 converged, 2 breakdown */
	LISHIP_KS_NOT_HALF,      /* BiCGSTAB: 0 only between the half-step convergence and its x update (every step re-arms it) */
	LISHIP_KS_NHIST,         /* residual-history entries written (a breakdown ends an iteration without one) */
	LISHIP_KS_LEN = 32
};
enum {
	LISHIP_STEP_CG_ALPHA = 1,     /* after <p,q>:            breakdown test, alpha = rho / <p,q>                */
	LISHIP_STEP_CG_RESID,         /* after the update pass:  ||r||, history, convergence, rho <- <r,r>, beta    */
	LISHIP_STEP_CG_RESID_PRE,     /* same with rho <- <r,z> (second sum)                                        */
	LISHIP_STEP_BICGSTAB_ALPHA,   /* after <rtld,v>: rho == 0 breakdown test, alpha = rho / <rtld,v>            */
	LISHIP_STEP_BICGSTAB_HALF,    /* after ||s||: convergence at the half step (lowers NOT_HALF for the x update)*/
	LISHIP_STEP_BICGSTAB_OMEGA,   /* after <t,s>,<t,t>: omega = <t,s>/<t,t>                                     */
	LISHIP_STEP_BICGSTAB_RESID,   /* after the r update: ||r||, history, convergence, omega breakdown, rho', beta */
	LISHIP_STEP_BICG_ALPHA,       /* after <p~,q>: rho == 0 and <p~,q> == 0 breakdown tests, alpha              */
	LISHIP_STEP_BICG_RESID,       /* after the x,r update: ||r||, history, convergence                          */
	LISHIP_STEP_BICG_RHO          /* after the r~ update: rho <- <r~, M^-1 r>, beta                             */
};
/* install (flag != NULL) or remove (NULL) the guard for the launches that follow */
int  liship_krylov_guard(const double *flag);
/* one scalar step on `state`; rhistory (device, may be NULL) receives the residual norm of each completed
 * iteration at [iter].  In a multi-rank job `gathered` holds nranks x count per-rank sums (rank-major), which are
 * added in rank order into the slots the step reads, before it runs; NULL otherwise. */
int  liship_krylov_step(int step, double *state, double *rhistory, const double *gathered, int nranks, void *stream);
/* single-rank jobs: announce the step BEFORE launching the reduction whose sums it reads; it then runs in that
 * reduction's last kernel instead of a launch of its own.  liship_krylov_chain_flush runs a step that was announced
 * but found no reduction to ride in (call it after the reduction; a no-op otherwise); step 0 withdraws an announced step. */
int  liship_krylov_chain(int step, double *state, double *rhistory);
/* out[k] = sum over ranks r (in rank order, from 0.0) of gathered[r*count + k]: the device half of the cross-rank
 * fold that MPI_Allreduce does in the reference (lis_vector_ops.c:119,263), count <= 64 */
int  liship_rank_fold_f64(int count, const double *gathered, int nranks, double *out, void *stream);
int  liship_krylov_chain_flush(void *stream);
/* y = x + (*pa)*y;  y = x.*d + (*pa)*y;  y += (*pa)*x;  y += (*pa)*x, then y += (*pb)*w;  y += (*pa)*x, then y = w + (*pb)*y */
int  liship_xpay_dev_f64(int n, const double *x, const double *pa, double *y, void *stream);
int  liship_pmul_xpay_dev_f64(int n, const double *x, const double *d, const double *pa, double *y, void *stream);
int  liship_axpy_dev_f64(int n, const double *pa, const double *x, double *y, void *stream);
int  liship_axpy2_dev_f64(int n, const double *pa, const double *x, const double *pb, const double *w, double *y, void *stream);
int  liship_axpy_xpay_dev_f64(int n, const double *pa, const double *x, const double *w, const double *pb, double *y, void *stream);
/* CG with the x update deferred into the NEXT iteration's direction update, which reads p anyway (lis_solver_cg.c:199, then
 * :176-183): x += (*palpha)*p on the old p, then p = r.*dinv + (*pbeta)*p (dinv NULL: p = r + (*pbeta)*p).  Each element sees
 * the reference's operations in the reference's order -- the bits of x and p are those of the undeferred loop -- while an
 * iteration moves 80 instead of 88 B of vectors per row.  palpha NULL: first iteration, x is not touched.  The caller owes
 * the last x += alpha*p after the loop (liship_axpy_dev_f64). */
int  liship_cg_direction_dev_f64(int n, const double *palpha, const double *pbeta, const double *r, const double *dinv,
                                 double *p, double *x, void *stream);
/* r += (*pna)*q (pna holds -alpha) ; result = {sum r^2, sum r*(r.*dinv)} : the residual half of liship_cg_update_jacobi_f64
 * (without a preconditioner liship_axpy_sumsq_dev_f64 serves) */
int  liship_cg_residual_jacobi_dev_f64(int n, const double *pna, const double *q, const double *dinv, double *r,
                                       double *result, void *work, void *stream);
/* The two passes for a Jacobi preconditioner with a UNIFORM diagonal: every dinv[i] is the double dc (the caller has made sure:
 * liship_count_ne_f64 returned 0), so r[i]*dc is r[i]*dinv[i] in every bit and the array is not read -- 64 instead of 80 B of
 * vectors per row and iteration. */
int  liship_cg_direction_uniform_dev_f64(int n, const double *palpha, const double *pbeta, const double *r, double dc,
                                         double *p, double *x, void *stream);
int  liship_cg_residual_jacobi_uniform_dev_f64(int n, const double *pna, const double *q, double dc, double *r,
                                               double *result, void *work, void *stream);
/* result[0] = how many x[i] differ from `a` in any bit (a reduction like the others: work as for liship_dot_f64) */
int  liship_count_ne_f64(int n, const double *x, double a, double *result, void *work, void *stream);
/* BiCGSTAB without a preconditioner (shat is s itself): x += (*palpha)*phat, x += (*pomega)*s, r = s + (*pnomega)*t, result =
 * {sum r^2, sum rtld*r} -- lis_solver_bicgstab.c:272-279 and :190 of the next iteration in ONE pass (s is read once for the
 * iterate and the residual: 56 instead of 64 B per row); s is the content of r[] on entry */
int  liship_bicgstab_end_dev_f64(int n, const double *palpha, const double *pomega, const double *pnomega, const double *phat,
                                 const double *t, const double *rtld, double *x, double *r, double *result, void *work, void *stream);
/* the fused update passes with the coefficient in HBM (dinv may be NULL for liship_cg_update_dev_f64) */
int  liship_cg_update_dev_f64(int n, const double *palpha, const double *p, const double *q, const double *dinv,
                              double *x, double *r, double *result, void *work, void *stream);
int  liship_axpy_sumsq_dev_f64(int n, const double *pa, const double *x, double *y, double *result, void *work, void *stream);
int  liship_axpy_sumsq_dot_dev_f64(int n, const double *pa, const double *x, double *y, const double *v, double *result,
                                   void *work, void *stream);
/* z = c0*v0, z += c1*v1, ... (accumulate = 0) or z += c0*v0, ... (accumulate = 1), element by element in that
 * order: the bits of lis_vector_scale/axpy chains (lis_solver_gmres.c:290-296, :323-329) in one pass over z.
 * vs[] and coef[] are HOST arrays (passed by value to the kernel); a v that aliases z reads z as it was on entry.
 * More than 48 vectors are processed in chunks: only the first chunk may then alias z. */
int  liship_lincomb_f64(int n, int count, const double *const *vs, const double *coef, int accumulate,
                        double *z, void *stream);
/* result[0] = <x,y>, result[1] = <x,x> in one pass (BiCGSTAB's <t,s>,<t,t>, lis_solver_bicgstab.c:267-268) */
int  liship_dot2_f64(int n, const double *x, const double *y, double *result, void *work, void *stream);

/* ------------------------------------------------------------------ matrix helpers on device */
/* d[i] = first value[j] with index[j]==i in row i, else 0: lis_matrix_get_diagonal_csr,
 * src/matrix/lis_matrix_csr.c:547-558 */
int  liship_csr_diagonal_f64(int n, const int *ptr, const int *index, const double *value,
                             double *d, void *stream);
/* XCD strips of the native ELL / DIA kernels (round 6): rows per plane of the structured grid the next whole-matrix ELL / DIA launches work on -- every XCD then takes
 * one eighth of every plane and walks the planes in order, so the +-plane neighbours of a row are in its own L2 (the CSR kernels learn their plane at plan time:
 * liship_csr_plan_strip_rows).  0: the natural workgroup order.  An order of the workgroups only: the bits cannot depend on it. */
int  liship_spmv_formats_set_plane(int rows);
/* the plane of a structured grid from ELL arrays in HBM: the largest |column - row| when at least half of the rows reach it (as liship_csr_plan_scan_band), else 0 */
int  liship_ell_scan_band(int n, int maxnzr, const int *index, int *plane_rows, void *stream);
/* the same for the native ELL arrays (first slot whose index is the row; padding gives 0: src/matrix/lis_matrix_ell.c lis_matrix_get_diagonal_ell) and the
 * native DIA arrays (the stored diagonal of offset 0, one chunk value[d*n + i]: src/matrix/lis_matrix_dia.c lis_matrix_get_diagonal_dia) */
int  liship_ell_diagonal_f64(int n, int maxnzr, const int *index, const double *value, double *d, void *stream);
int  liship_dia_diagonal_f64(int n, int nnd, const int *offsets, const double *value, double *d, void *stream);
/* halo pack: ws[i] = x[export_index[i]]  (lis_send_recv, src/matrix/lis_matrix_mpi.c:904-916) */
int  liship_gather_f64(int count, const int *export_index, const double *x, double *ws, void *stream);
/* A^T of a CSR matrix in HBM, each transposed row listing its entries in the order of their positions in the
 * source arrays -- the order lis_matvech_csr's scatter adds them (lis_matvec_csr.c:213-250).  tptr: ncols+1 ints,
 * tindex / tvalue: nnz entries, work: ncols + nnz ints.  Setup-time (once per matrix). */
/* y[0..rows) = A^T x from the transposed CSR below (rows = columns of A, nsrc = rows of A), summed as the OpenMP build of lis_matvech_csr sums with
 * T threads: per thread chunk LIS_GET_ISIE(k, T, nsrc) of SOURCE rows left to right from 0.0, chunk sums in chunk order (src/matvec/lis_matvec_csr.c:
 * 207-236).  The reference-order mode's A^T x for T > 1 (T = 1 is the plain row sum). */
int  liship_spmv_csr_transposed_chunked_f64(int rows, int nsrc, int T, const int *tptr, const int *tidx, const double *tval,
                                            const double *x, double *y, void *stream);
int  liship_csr_transpose_f64(int nrows, int ncols, int nnz, const int *ptr, const int *index, const double *value,
                              int *tptr, int *tindex, double *tvalue, int *work, void *stream);
/* ---- storage-format conversions of a CSR matrix that lives in HBM (kernels/convert.hip): the arrays the reference's host routines
 * build (src/matrix/lis_matrix_ell.c:958-1070, lis_matrix_dia.c:1191-1304, lis_matrix_bsr.c:351-552), bit for bit, from device arrays.
 * liship_csr_row_facts: facts[0] = longest row, facts[1] = 1 when some row is not in ascending column order (two device ints).
 * ELL: value 0 on the row's own column pads short rows.  DIA (rows in ascending column order): liship_csr_dia_offsets marks the
 * offsets that occur (used: n + ncols ints; slot: n + ncols + 1; scratch: (n + ncols) / 4096 + 4 long long) and returns their number,
 * liship_csr_to_dia writes them ascending and value[d * n + i].  BSR: distinct block columns of a block row in first-seen order, blocks
 * column-major; liship_csr_bsr_count gives bptr and the block count (-1: a block row with more than 96 distinct blocks; count: nr + 1
 * ints, scratch: nr / 4096 + 4 long long).  The *_rows forms build the CSR row form of an ELL / DIA matrix (lis_device.c): the terms of
 * a row in the format's own order, padding and explicit zeros included. */
int  liship_csr_row_facts(int n, const int *ptr, const int *index, int *facts, void *stream);
int  liship_csr_to_ell(int n, int maxnzr, const int *ptr, const int *index, const double *value, int *ell_index, double *ell_value, void *stream);
/* the row form of a BSR matrix (as liship_csr_to_ell_rows for ELL): CSR rows that list lis_matvec_bsr's terms of each scalar row -- block after block, column after column,
 * explicit zeros included -- so that the CSR kernels form the same sums and the value records apply to constant-coefficient block matrices */
int  liship_bsr_to_rows(int n, int bnr, int bnc, const int *bptr, const int *bindex, const double *value, int *rptr, int *rindex, double *rvalue, void *stream);
int  liship_csr_to_ell_rows(int n, int maxnzr, const int *ptr, const int *index, const double *value, int *rptr, int *rindex, double *rvalue, void *stream);
int  liship_csr_dia_offsets(int n, int ncols, const int *ptr, const int *index, int *used, int *slot, long long *scratch, int *nnd, void *stream);
int  liship_csr_to_dia(int n, int ncols, int nnd, const int *ptr, const int *index, const double *value, const int *used, const int *slot,
                       int *offsets, double *dia_value, void *stream);
int  liship_dia_row_counts(int n, int ncols, int nnd, const int *offsets, int *count, int *rptr, long long *scratch, int *rnnz, void *stream);
int  liship_dia_to_rows(int n, int ncols, int nnd, const int *offsets, const double *dia_value, const int *rptr, int *rindex, double *rvalue, void *stream);
int  liship_csr_bsr_count(int n, int np, int bnr, int bnc, const int *ptr, const int *index, int *count, int *bptr, long long *scratch, int *bnnz, void *stream);
int  liship_csr_to_bsr(int n, int bnr, int bnc, int bnnz, const int *ptr, const int *index, const double *value, const int *bptr,
                       int *bindex, double *bsr_value, void *stream);
/* JAD from the length-sorted row order perm[n] and the jagged-diagonal starts jptr[maxnzr + 1] (both made on the host: the order is the
 * reference's unstable quicksort, lis_convert.c): entry j of row perm[s] goes to [jptr[j] + s] */
int  liship_csr_to_jad(int n, const int *perm, const int *jptr, const int *ptr, const int *index, const double *value,
                       int *jad_index, double *jad_value, void *stream);
/* reverse halo: y[export_index[i]] += wr[i], indices unique within one call  (lis_reduce, lis_matrix_mpi.c:988-996) */
int  liship_scatter_add_f64(int count, const int *export_index, const double *wr, double *y, void *stream);

/* ------------------------------------------------------------------ synthetic inputs
 * Rows [is,ie) of the 3-D 7-pt Poisson matrix on an l x m x n grid, generated directly in HBM with the
 * entry order of test/test3.c:114-127 (sorted!=0: ascending column as test/spmvtest3.c:192-195).
 * Columns are LOCAL: owned column g -> g-is, ghost columns -> (ie-is) + rank in ascending global order,
 * exactly as lis_matrix_g2l_csr numbers them (src/matrix/lis_matrix_mpi.c:274-306).
 * ptr has ie-is+1 entries; index/value must hold liship_poisson3d_nnz() entries. */
long long liship_poisson3d_nnz(int l, int m, int n, int is, int ie);
int  liship_poisson3d_csr(int l, int m, int n, int is, int ie, int sorted,
                          int *ptr, int *index, double *value, void *stream);
/* b = A*1 for those rows without forming A (test/test3.c:150): 6 minus the number of neighbours */
int  liship_poisson3d_rhs(int l, int m, int n, int is, int ie, double *b, void *stream);

#ifdef __cplusplus
}
#endif
#endif
