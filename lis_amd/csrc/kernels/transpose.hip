// transpose.hip -- A^T of a CSR matrix, built in HBM, in the order lis_matvech_csr needs.
//
// lis_matvech_csr (reference src/matvec/lis_matvec_csr.c:213-250) scatters y[col] += a * x[row] while it walks
// the matrix row by row, so the contributions to y[c] arrive ordered by their position k in the CSR arrays.
// The transposed operator used by lis_matvech.c must list them in that order.  Three steps, no sort library:
//   1. count entries per column (integer atomics: order-free), exclusive scan -> tptr
//   2. scatter every entry's POSITION k into its column's segment in whatever order the atomics grant
//   3. one lane per column sorts its segment of positions ascending (positions are unique, so the result does
//      not depend on step 2's order), then fills index = row_of(k) (binary search in ptr) and value = value[k]
// Setup-time code: runs once per matrix, not on the hot path.
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK)
void count_columns(int n, const int *__restrict__ ptr, const int *__restrict__ idx, int *__restrict__ count)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    for (int k = ptr[r]; k < ptr[r + 1]; k++) atomicAdd(&count[idx[k]], 1);
}

// exclusive scan of count[0..m) into out[0..m] in three passes: sums of 4096-element tiles, a single workgroup
// scanning the tile sums, tiles scanned locally with their offset
constexpr int TILE = 4096;

__global__ __launch_bounds__(BLOCK)
void tile_sums(int m, const int *__restrict__ count, long long *__restrict__ sums)
{
    __shared__ long long part[BLOCK];
    const long long base = (long long)blockIdx.x * TILE;
    long long s = 0;
    for (int i = threadIdx.x; i < TILE; i += BLOCK) if (base + i < m) s += count[base + i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1) { if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(1024)
void scan_tile_sums(int ntiles, long long *__restrict__ sums)      // in place: sums[t] = sum of tiles before t
{
    __shared__ long long part[1024];
    const int t = threadIdx.x, T = blockDim.x;
    const int per = (ntiles + T - 1) / T;
    const int lo = min(ntiles, t * per), hi = min(ntiles, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += sums[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { long long run = 0; for (int i = 0; i < T; i++) { const long long v = part[i]; part[i] = run; run += v; } }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; i++) { const long long v = sums[i]; sums[i] = run; run += v; }
}

__global__ __launch_bounds__(BLOCK)
void scan_tiles(int m, const int *__restrict__ count, const long long *__restrict__ sums, int *__restrict__ out)
{
    __shared__ long long part[BLOCK];
    constexpr int PER = TILE / BLOCK;                 // consecutive elements per lane
    const long long base = (long long)blockIdx.x * TILE + (long long)threadIdx.x * PER;
    long long s = 0;
    for (int i = 0; i < PER; i++) if (base + i < m) s += count[base + i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { long long run = sums[blockIdx.x]; for (int i = 0; i < BLOCK; i++) { const long long v = part[i]; part[i] = run; run += v; } }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = 0; i < PER; i++) if (base + i < m) { out[base + i] = (int)run; run += count[base + i]; }
    if (base <= m && m < base + PER) out[m] = (int)run;              // the total, by the lane whose range holds m
    if (m % TILE == 0 && blockIdx.x == gridDim.x - 1 && threadIdx.x == BLOCK - 1) out[m] = (int)run;
}

__global__ __launch_bounds__(BLOCK)
void scatter_positions(int n, const int *__restrict__ ptr, const int *__restrict__ idx,
                       const int *__restrict__ tptr, int *__restrict__ fill, int *__restrict__ pos, int *__restrict__ rws)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    for (int k = ptr[r]; k < ptr[r + 1]; k++) {
        const int c = idx[k];
        const int at = tptr[c] + atomicAdd(&fill[c], 1);
        pos[at] = k;
        if (rws) rws[at] = r;                         // the row travels with the position (round 6): no search for it afterwards
    }
}

// (a, b): positions and, when b is not NULL, their rows -- sorted together by position
__device__ void sift_down(int *a, int *b, int start, int end)
{
    int root = start;
    while (2 * root + 1 <= end) {
        int child = 2 * root + 1;
        if (child + 1 <= end && a[child] < a[child + 1]) child++;
        if (a[root] >= a[child]) return;
        const int t = a[root]; a[root] = a[child]; a[child] = t;
        if (b) { const int u = b[root]; b[root] = b[child]; b[child] = u; }
        root = child;
    }
}

constexpr int MID_FROM = 33, MID_TO = 2048;         // segment lengths a wavefront sorts by ranks in LDS (order_mid)

// one lane per column: segments of up to 32 positions (insertion sort) and those beyond MID_TO (heap sort in place); the lengths in between are order_mid's
// when `mid` says so.  rws == NULL (no room for the rows): the row of position k by binary search in ptr, as rounds 2-5 did for every entry.
__global__ __launch_bounds__(BLOCK)
void order_and_fill(int ncols, int nrows, const int *__restrict__ ptr, const double *__restrict__ val,
                    const int *__restrict__ tptr, int *__restrict__ pos, int *__restrict__ rws, int *__restrict__ tidx,
                    double *__restrict__ tval, int mid)
{
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= ncols) return;
    int *a = pos + tptr[c], *b = rws ? rws + tptr[c] : nullptr;
    const int len = tptr[c + 1] - tptr[c];
    if (mid && len >= MID_FROM && len <= MID_TO) return;
    if (len <= 32) {                                 // insertion sort
        for (int i = 1; i < len; i++) {
            const int v = a[i], w = b ? b[i] : 0;
            int j = i - 1;
            while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; if (b) b[j + 1] = b[j]; j--; }
            a[j + 1] = v;
            if (b) b[j + 1] = w;
        }
    } else {                                         // heap sort, in place
        for (int s = (len - 2) / 2; s >= 0; s--) sift_down(a, b, s, len - 1);
        for (int e = len - 1; e > 0; e--) {
            const int t = a[e]; a[e] = a[0]; a[0] = t;
            if (b) { const int u = b[e]; b[e] = b[0]; b[0] = u; }
            sift_down(a, b, 0, e - 1);
        }
    }
    for (int i = 0; i < len; i++) {
        const int k = a[i];
        int lo = 0;
        if (b) lo = b[i];
        else {
            int hi = nrows;                          // last row r with ptr[r] <= k
            while (hi - lo > 1) { const int m2 = lo + ((hi - lo) >> 1); if (ptr[m2] <= k) lo = m2; else hi = m2; }
        }
        tidx[tptr[c] + i] = lo;
        tval[tptr[c] + i] = val[k];
    }
}

// one WAVEFRONT per column, segments of MID_FROM .. MID_TO positions (the columns of a finite-element matrix: 81 entries; round 6): the segment goes to LDS, every
// lane finds the rank of its elements by counting the smaller positions (positions are unique), and the entry lands where its rank says -- coalesced loads, no
// dependent chain through global memory.  The lane-per-column heap sort took 0.23 s on the Queen-class matrix (most of the first BiCG solve).
__global__ __launch_bounds__(BLOCK)
void order_mid(int ncols, const double *__restrict__ val, const int *__restrict__ tptr, const int *__restrict__ pos, const int *__restrict__ rws,
               int *__restrict__ tidx, double *__restrict__ tval)
{
    __shared__ int keys[BLOCK / WAVE][MID_TO];
    const int wave = (int)threadIdx.x / WAVE, lane = (int)threadIdx.x & (WAVE - 1);
    const int c = blockIdx.x * (BLOCK / WAVE) + wave;
    const int b = c < ncols ? tptr[c] : 0, len = c < ncols ? tptr[c + 1] - b : 0;
    const bool mine_to_do = len >= MID_FROM && len <= MID_TO;
    int *k = keys[wave];
    if (mine_to_do) for (int i = lane; i < len; i += WAVE) k[i] = pos[b + i];
    __syncthreads();                                  // (every wavefront of the workgroup gets here, with or without a column of its own)
    if (!mine_to_do) return;
    for (int i = lane; i < len; i += WAVE) {
        const int mine = k[i];
        int rank = 0;
        for (int j = 0; j < len; j++) rank += k[j] < mine;
        tidx[b + rank] = rws[b + i];
        tval[b + rank] = val[mine];
    }
}

} // namespace

// tptr: ncols + 1 ints, tidx / tval: nnz entries; work: (ncols + nnz) ints of scratch.  All device pointers.
extern "C" int liship_csr_transpose_f64(int nrows, int ncols, int nnz, const int *ptr, const int *idx, const double *val,
                                        int *tptr, int *tidx, double *tval, int *work, void *stream)
{
    if (nrows < 0 || ncols < 0 || nnz < 0) return LISHIP_ERR_ARG;
    hipStream_t st = as_stream(stream);
    int *count = work, *pos = work + ncols;
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * (size_t)(ncols > 0 ? ncols : 1), st));
    if (nrows > 0) count_columns<<<(nrows + BLOCK - 1) / BLOCK, BLOCK, 0, st>>>(nrows, ptr, idx, count);
    LAUNCH_CHECK();
    {
        const int ntiles = (ncols + TILE - 1) / TILE;
        long long *sums = nullptr;
        HIP_TRY(hipMalloc(&sums, sizeof(long long) * (size_t)(ntiles > 0 ? ntiles : 1)));
        if (ntiles > 0) {
            tile_sums<<<ntiles, BLOCK, 0, st>>>(ncols, count, sums);
            scan_tile_sums<<<1, 1024, 0, st>>>(ntiles, sums);
            scan_tiles<<<ntiles, BLOCK, 0, st>>>(ncols, count, sums, tptr);
        } else HIP_TRY(hipMemsetAsync(tptr, 0, sizeof(int), st));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(sums);
        if (e != hipSuccess) return (int)e;
    }
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * (size_t)(ncols > 0 ? ncols : 1), st));
    int *rws = nullptr;                              // the row of every scattered position (nnz ints of scratch of its own; without it: the binary searches of rounds 2-5)
    if (nnz > 0 && hipMalloc(&rws, sizeof(int) * (size_t)nnz) != hipSuccess) { (void)hipGetLastError(); rws = nullptr; }
    if (nrows > 0) scatter_positions<<<(nrows + BLOCK - 1) / BLOCK, BLOCK, 0, st>>>(nrows, ptr, idx, tptr, count, pos, rws);
    hipError_t e = hipGetLastError();
    const bool mid = rws != nullptr && nrows > 0 && (long long)nnz > 8ll * ncols;      // (short rows on average: no column worth a wavefront)
    if (e == hipSuccess && ncols > 0) { order_and_fill<<<(ncols + BLOCK - 1) / BLOCK, BLOCK, 0, st>>>(ncols, nrows, ptr, val, tptr, pos, rws, tidx, tval, mid ? 1 : 0); e = hipGetLastError(); }
    if (e == hipSuccess && mid) { order_mid<<<(ncols + BLOCK / WAVE - 1) / (BLOCK / WAVE), BLOCK, 0, st>>>(ncols, val, tptr, pos, rws, tidx, tval); e = hipGetLastError(); }
    if (rws) { if (e == hipSuccess) e = hipStreamSynchronize(st); (void)hipFree(rws); }
    return e == hipSuccess ? 0 : (int)e;
}

// ---- A^T x in the reference's order for OMP_NUM_THREADS = T > 1 (parity mode, liship_set_reference_reductions) ----------------
// The OpenMP build of lis_matvech_csr gives thread k the source rows LIS_GET_ISIE(k, T, n), lets it scatter into a buffer of its own
// from 0.0, and then forms y[c] = ((0.0 + w[0][c]) + w[1][c]) + ... (src/matvec/lis_matvec_csr.c:207-236).  On the transposed rows
// (entries of row c in ascending source row = the scatter order) that is: the entries of one thread's chunk summed left to right
// from 0.0, the chunk sums added in chunk order from 0.0.  One lane per row: a parity mode, not a fast one.
namespace {
__global__ __launch_bounds__(BLOCK)
void spmv_t_chunked_kernel(int rows, int nsrc, int T, const int *__restrict__ tptr, const int *__restrict__ tidx,
                           const double *__restrict__ tval, const double *__restrict__ x, double *__restrict__ y, const double *skip)
{
    if (skip && skip[0] != 0.0) return;
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= rows) return;
    const int q = nsrc / T, rem = nsrc % T;
    const long long big = (long long)rem * (q + 1);      // rows below `big` lie in chunks of q + 1 rows
    double total = 0.0, part = 0.0;
    int chunk = 0;
    for (int k = tptr[c]; k < tptr[c + 1]; k++) {
        const int j = tidx[k];
        const int kc = j < big ? j / (q + 1) : rem + (int)((j - big) / q);
        if (kc != chunk) { total += part; part = 0.0; chunk = kc; }      // (adding a chunk without entries adds +0.0: nothing)
        part += tval[k] * x[j];
    }
    total += part;
    y[c] = total;
}
} // namespace

// y[0..rows) = A^T x from the transposed CSR (rows = columns of A, nsrc = rows of A), summed as the reference's T threads do
extern "C" int liship_spmv_csr_transposed_chunked_f64(int rows, int nsrc, int T, const int *tptr, const int *tidx, const double *tval,
                                                      const double *x, double *y, void *stream)
{
    if (rows < 0 || nsrc < 0 || T < 1) return LISHIP_ERR_ARG;
    if (rows == 0) return 0;
    spmv_t_chunked_kernel<<<(rows + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(stream)>>>(rows, nsrc, T, tptr, tidx, tval, x, y, liship_internal_guard());
    LAUNCH_CHECK();
    return 0;
}
