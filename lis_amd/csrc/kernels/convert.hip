// convert.hip -- storage-format conversions of a CSR matrix that already lives in HBM (gfx950).
//
// The reference converts on the host (src/matrix/lis_matrix_ell.c:958-1070 csr2ell, lis_matrix_dia.c:1191-1304 csr2dia,
// lis_matrix_csc.c:904-1087 csr2csc, lis_matrix_bsr.c:351-552 csr2bsr); its own drivers convert once per format per run
// (test/spmvtest3.c:214-218).  These kernels produce the SAME arrays -- same padding, same diagonal order, same block order, bit for bit
// (the host versions in lis_convert.c are the checker: tests compare against the reference-made goldens either way) -- from the HBM copy,
// so a conversion costs a pass over HBM instead of a pass over host memory plus an upload.  Integer work; setup-time, not the hot path.
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;
inline int grid_for(long long n) { return (int)((n + BLOCK - 1) / BLOCK); }

// ---- exclusive scan of count[0..m) into out[0..m] (out[m] = total; 64-bit tile sums): tile sums, one workgroup over them, tiles
constexpr int TILE = 4096;
__global__ __launch_bounds__(BLOCK)
void scan_tile_sums(int m, const int *__restrict__ count, long long *__restrict__ sums)
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
__global__ __launch_bounds__(1024) void scan_tile_offsets(int ntiles, long long *__restrict__ sums)      // one workgroup: sums[t] = sum of the tiles before t, sums[ntiles] = all
{                                                                                                       // (one THREAD walked the 32 768 tiles of a 512^3 conversion in 2.3 ms)
    __shared__ long long part[1024];
    const int t = threadIdx.x, per = (ntiles + 1023) / 1024;
    const int lo = min(ntiles, t * per), hi = min(ntiles, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += sums[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { long long run = 0; for (int i = 0; i < 1024; i++) { const long long v = part[i]; part[i] = run; run += v; } sums[ntiles] = run; }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; i++) { const long long v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(BLOCK)
void scan_tiles(int m, const int *__restrict__ count, const long long *__restrict__ sums, int *__restrict__ out)
{
    __shared__ int part[BLOCK];
    const long long base = (long long)blockIdx.x * TILE;
    constexpr int PER = TILE / BLOCK;
    int mine[PER], s = 0;
    for (int u = 0; u < PER; u++) { const long long i = base + (long long)threadIdx.x * PER + u; mine[u] = i < m ? count[i] : 0; s += mine[u]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < BLOCK; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = sums[blockIdx.x] + part[threadIdx.x] - s;
    for (int u = 0; u < PER; u++) { const long long i = base + (long long)threadIdx.x * PER + u; if (i < m) out[i] = (int)run; run += mine[u]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == BLOCK - 1) out[m] = (int)sums[gridDim.x];
}
// out[0..m]; scratch: (tiles + 1) long long
int exclusive_scan(int m, const int *count, int *out, long long *scratch, hipStream_t st)
{
    const int tiles = (m + TILE - 1) / TILE;
    if (m <= 0) { HIP_TRY(hipMemsetAsync(out, 0, sizeof(int), st)); return 0; }
    scan_tile_sums<<<tiles, BLOCK, 0, st>>>(m, count, scratch);
    scan_tile_offsets<<<1, 1024, 0, st>>>(tiles, scratch);
    scan_tiles<<<tiles, BLOCK, 0, st>>>(m, count, scratch, out);
    LAUNCH_CHECK();
    return 0;
}

// ---- facts about the CSR rows: longest row, whether every row lists its columns in ascending order
__global__ __launch_bounds__(BLOCK)
void csr_row_facts(int n, const int *__restrict__ ptr, const int *__restrict__ idx, int *__restrict__ facts)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    int len = 0, unsorted = 0;
    if (r < n) {
        const int s = ptr[r], e = ptr[r + 1];
        len = e - s;
        for (int k = s + 1; k < e; k++) unsorted |= idx[k] < idx[k - 1];
    }
    __shared__ int mx[BLOCK], us[BLOCK];
    mx[threadIdx.x] = len; us[threadIdx.x] = unsorted;
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { mx[threadIdx.x] = max(mx[threadIdx.x], mx[threadIdx.x + w]); us[threadIdx.x] |= us[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicMax(&facts[0], mx[0]); if (us[0]) atomicOr(&facts[1], 1); }
}

// ---- ELL (lis_matrix_ell.c:1020-1045): slot j of row i at [j*n + i]; padding: value 0 on the row's own column
__global__ __launch_bounds__(BLOCK)
void csr_to_ell(int n, int maxnzr, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                int *__restrict__ eidx, double *__restrict__ eval)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int s = ptr[i], len = ptr[i + 1] - s;
    for (int j = 0; j < maxnzr; j++) {
        const bool have = j < len;
        eidx[(size_t)j * n + i] = have ? idx[s + j] : i;
        eval[(size_t)j * n + i] = have ? val[s + j] : 0.0;
    }
}
// the row form of the same matrix (lis_device.c try_row_form): CSR rows of exactly maxnzr terms, padding included
__global__ __launch_bounds__(BLOCK)
void csr_to_ell_rows(int n, int maxnzr, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                     int *__restrict__ rptr, int *__restrict__ ridx, double *__restrict__ rval)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n) return;
    rptr[i] = i * maxnzr;
    if (i == n) return;
    const int s = ptr[i], len = ptr[i + 1] - s;
    for (int j = 0; j < maxnzr; j++) {
        const bool have = j < len;
        ridx[(size_t)i * maxnzr + j] = have ? idx[s + j] : i;
        rval[(size_t)i * maxnzr + j] = have ? val[s + j] : 0.0;
    }
}

// ---- DIA (lis_matrix_dia.c:1224-1300; rows already in ascending column order): the offsets that occur, ascending; value[d*n + i]
__global__ __launch_bounds__(BLOCK)
void dia_mark(int n, const int *__restrict__ ptr, const int *__restrict__ idx, int *__restrict__ used)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int k = ptr[i]; k < ptr[i + 1]; k++) { const int o = idx[k] - i + n; if (!used[o]) used[o] = 1; }     // racing writers store the same 1
}
__global__ __launch_bounds__(BLOCK)
void dia_offsets(int span, int n, const int *__restrict__ used, const int *__restrict__ slot, int *__restrict__ offs)
{
    const int o = blockIdx.x * BLOCK + threadIdx.x;
    if (o < span && used[o]) offs[slot[o]] = o - n;
}
__global__ __launch_bounds__(BLOCK)
void csr_to_dia(int n, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                const int *__restrict__ slot, double *__restrict__ dval)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int k = ptr[i]; k < ptr[i + 1]; k++) dval[(size_t)slot[idx[k] - i + n] * n + i] = val[k];
}
// row form: the diagonals that reach row i (0 <= i + offset < ncols), ascending, explicit zeros included (lis_matvec_dia.c:154-160)
__global__ __launch_bounds__(BLOCK)
void dia_row_counts(int n, int ncols, int nnd, const int *__restrict__ offs, int *__restrict__ count)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int c = 0;
    for (int d = 0; d < nnd; d++) { const long long j = (long long)i + offs[d]; c += (j >= 0 && j < ncols); }
    count[i] = c;
}
__global__ __launch_bounds__(BLOCK)
void dia_to_rows(int n, int ncols, int nnd, const int *__restrict__ offs, const double *__restrict__ dval,
                 const int *__restrict__ rptr, int *__restrict__ ridx, double *__restrict__ rval)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int at = rptr[i];
    for (int d = 0; d < nnd; d++) {
        const long long j = (long long)i + offs[d];
        if (j >= 0 && j < ncols) { ridx[at] = (int)j; rval[at] = dval[(size_t)d * n + i]; at++; }
    }
}

// ---- BSR (lis_matrix_bsr.c:351-552): the distinct block columns of a block row in FIRST-SEEN order (rows of the block row top to
// bottom, entries left to right), blocks column-major, explicit zeros; ghost columns start on a fresh block column (:425-428)
constexpr int BSR_LIST = 96;          // distinct blocks of one block row a lane keeps (more: the host converts)
__device__ __forceinline__ int bcol_of(int c, int n, int pad, int bnc) { return ((c < n ? c : c + pad) / bnc); }
__device__ __forceinline__ int boff_of(int c, int n, int pad, int bnc) { return ((c < n ? c : c + pad) % bnc); }
__global__ __launch_bounds__(BLOCK)
void bsr_count(int n, int nr, int bnr, int bnc, int pad, const int *__restrict__ ptr, const int *__restrict__ idx,
               int *__restrict__ count, int *__restrict__ overflow)
{
    const int br = blockIdx.x * BLOCK + threadIdx.x;
    if (br >= nr) return;
    int list[BSR_LIST], seen = 0;
    for (int ii = 0; ii < bnr && br * bnr + ii < n; ii++)
        for (int k = ptr[br * bnr + ii]; k < ptr[br * bnr + ii + 1]; k++) {
            const int bc = bcol_of(idx[k], n, pad, bnc);
            int s = seen - 1;
            while (s >= 0 && list[s] != bc) s--;
            if (s < 0) { if (seen == BSR_LIST) { atomicOr(overflow, 1); count[br] = 0; return; } list[seen++] = bc; }
        }
    count[br] = seen;
}
__global__ __launch_bounds__(BLOCK)
void bsr_fill(int n, int nr, int bnr, int bnc, int pad, const int *__restrict__ ptr, const int *__restrict__ idx,
              const double *__restrict__ val, const int *__restrict__ bptr, int *__restrict__ bindex, double *__restrict__ bval)
{
    const int br = blockIdx.x * BLOCK + threadIdx.x;
    if (br >= nr) return;
    const int first = bptr[br], bs = bnr * bnc;
    int next = first;
    for (int ii = 0; ii < bnr && br * bnr + ii < n; ii++)
        for (int k = ptr[br * bnr + ii]; k < ptr[br * bnr + ii + 1]; k++) {
            const int bc = bcol_of(idx[k], n, pad, bnc), jc = boff_of(idx[k], n, pad, bnc);
            int s = next - 1;
            while (s >= first && bindex[s] != bc) s--;
            if (s < first) { s = next++; bindex[s] = bc; }              // (the value array was zeroed before the launch)
            bval[(size_t)s * bs + (size_t)jc * bnr + ii] = val[k];
        }
}

// ---- JAD (lis_matrix_jad.c:1690-1770, one chunk): slot s of the length-sorted order is row perm[s]; jagged diagonal j holds the j-th entry
// of every row long enough, at [jptr[j] + s]
__global__ __launch_bounds__(BLOCK)
void csr_to_jad(int n, const int *__restrict__ perm, const int *__restrict__ jptr, const int *__restrict__ ptr, const int *__restrict__ idx,
                const double *__restrict__ val, int *__restrict__ jidx, double *__restrict__ jval)
{
    const int s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    const int r = perm[s], src = ptr[r], cnt = ptr[r + 1] - src;
    for (int j = 0; j < cnt; j++) { jidx[jptr[j] + s] = idx[src + j]; jval[jptr[j] + s] = val[src + j]; }
}

} // namespace

// facts[0] = longest row, facts[1] = 1 when some row is not in ascending column order (device ints, zeroed here)
extern "C" int liship_csr_row_facts(int n, const int *ptr, const int *idx, int *facts, void *stream)
{
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemsetAsync(facts, 0, 2 * sizeof(int), st));
    if (n > 0) { csr_row_facts<<<grid_for(n), BLOCK, 0, st>>>(n, ptr, idx, facts); LAUNCH_CHECK(); }
    return 0;
}

extern "C" int liship_csr_to_ell(int n, int maxnzr, const int *ptr, const int *idx, const double *val, int *eidx, double *eval, void *stream)
{
    if (n > 0 && maxnzr > 0) { csr_to_ell<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, ptr, idx, val, eidx, eval); LAUNCH_CHECK(); }
    return 0;
}

namespace {
// the row form of a BSR matrix: scalar row r = bi * bnr + i lists, block after block of block row bi and column after column of the block, the terms
// lis_matvec_bsr adds to y[r] -- value[bc * bs + j * bnr + i] * x[bindex[bc] * bnc + j], explicit zeros included -- in that order (src/matvec/
// lis_matvec_bsr.c:57-150 generic, :293-343 2x2 ...).  Rows of one block row have the same length, so rptr is a closed form of bptr.
__global__ __launch_bounds__(BLOCK)
void bsr_to_rows(int n, int bnr, int bnc, const int *__restrict__ bptr, const int *__restrict__ bidx, const double *__restrict__ val,
                 int *__restrict__ rptr, int *__restrict__ ridx, double *__restrict__ rval)
{
    const long long r = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (r > n) return;
    if (r == n) {                                   // rptr[n]: everything the n rows hold (the padding rows of the last block row are not rows of A)
        const int bi = (n - 1) / bnr, i = n - bi * bnr;          // i = rows of the last block row that exist (1..bnr)
        rptr[n] = bnc * (bnr * bptr[bi] + i * (bptr[bi + 1] - bptr[bi]));
        return;
    }
    const int bi = (int)(r / bnr), i = (int)(r - (long long)bi * bnr);
    const int b0 = bptr[bi], nbk = bptr[bi + 1] - b0, bs = bnr * bnc;
    int at = bnc * (bnr * b0 + i * nbk);
    rptr[r] = at;
    for (int bc = b0; bc < b0 + nbk; bc++) {
        const int c0 = bidx[bc] * bnc;
        for (int j = 0; j < bnc; j++, at++) { ridx[at] = c0 + j; rval[at] = val[(size_t)bc * bs + (size_t)j * bnr + i]; }
    }
}
} // namespace

// rptr: n + 1 ints; ridx / rval: bnnz * bnr * bnc entries (fewer are used when the last block row is padded).  All device pointers.
extern "C" int liship_bsr_to_rows(int n, int bnr, int bnc, const int *bptr, const int *bindex, const double *value, int *rptr, int *rindex, double *rvalue, void *stream)
{
    if (n <= 0 || bnr < 1 || bnc < 1) return LISHIP_ERR_ARG;
    bsr_to_rows<<<grid_for((long long)n + 1), BLOCK, 0, as_stream(stream)>>>(n, bnr, bnc, bptr, bindex, value, rptr, rindex, rvalue);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_csr_to_ell_rows(int n, int maxnzr, const int *ptr, const int *idx, const double *val, int *rptr, int *ridx, double *rval, void *stream)
{
    csr_to_ell_rows<<<grid_for((long long)n + 1), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, ptr, idx, val, rptr, ridx, rval);
    LAUNCH_CHECK();
    return 0;
}

// DIA in two steps.  1: used[n + ncols] (ints, zeroed here) marks the offsets that occur, slot[] (n + ncols + 1 ints) is its exclusive scan,
// *nnd comes back to the host.  scratch: ((n + ncols) / 4096 + 2) long long.
extern "C" int liship_csr_dia_offsets(int n, int ncols, const int *ptr, const int *idx, int *used, int *slot, long long *scratch, int *nnd, void *stream)
{
    hipStream_t st = as_stream(stream);
    const int span = n + ncols;
    HIP_TRY(hipMemsetAsync(used, 0, sizeof(int) * (size_t)span, st));
    if (n > 0) { dia_mark<<<grid_for(n), BLOCK, 0, st>>>(n, ptr, idx, used); LAUNCH_CHECK(); }
    const int rc = exclusive_scan(span, used, slot, scratch, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(nnd, slot + span, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}
// 2: offs[nnd] ascending and value[nnd * n] (zero where a diagonal has no entry)
extern "C" int liship_csr_to_dia(int n, int ncols, int nnd, const int *ptr, const int *idx, const double *val, const int *used, const int *slot,
                                 int *offs, double *dval, void *stream)
{
    hipStream_t st = as_stream(stream);
    const int span = n + ncols;
    if (nnd > 0) {
        dia_offsets<<<grid_for(span), BLOCK, 0, st>>>(span, n, used, slot, offs);
        HIP_TRY(hipMemsetAsync(dval, 0, sizeof(double) * (size_t)n * (size_t)nnd, st));
        csr_to_dia<<<grid_for(n), BLOCK, 0, st>>>(n, ptr, idx, val, slot, dval);
        LAUNCH_CHECK();
    }
    return 0;
}
// the row form of a DIA matrix: rptr[n + 1] by a scan of the per-row counts (count: n ints, scratch: (n / 4096 + 2) long long); *rnnz to the host
extern "C" int liship_dia_row_counts(int n, int ncols, int nnd, const int *offs, int *count, int *rptr, long long *scratch, int *rnnz, void *stream)
{
    hipStream_t st = as_stream(stream);
    if (n > 0) { dia_row_counts<<<grid_for(n), BLOCK, 0, st>>>(n, ncols, nnd, offs, count); LAUNCH_CHECK(); }
    const int rc = exclusive_scan(n, count, rptr, scratch, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rnnz, rptr + n, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}
extern "C" int liship_dia_to_rows(int n, int ncols, int nnd, const int *offs, const double *dval, const int *rptr, int *ridx, double *rval, void *stream)
{
    if (n > 0) { dia_to_rows<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, offs, dval, rptr, ridx, rval); LAUNCH_CHECK(); }
    return 0;
}

// BSR in two steps.  1: bptr[nr + 1]; *bnnz to the host, -1 when a block row has more than 96 distinct blocks (the host converts then).
// count: nr + 1 ints (the last one is the overflow flag), scratch: (nr / 4096 + 2) long long
extern "C" int liship_csr_bsr_count(int n, int np, int bnr, int bnc, const int *ptr, const int *idx, int *count, int *bptr, long long *scratch, int *bnnz, void *stream)
{
    hipStream_t st = as_stream(stream);
    const int nr = n > 0 ? 1 + (n - 1) / bnr : 0, pad = (bnc - n % bnc) % bnc;
    (void)np;
    HIP_TRY(hipMemsetAsync(count + nr, 0, sizeof(int), st));
    if (nr > 0) { bsr_count<<<grid_for(nr), BLOCK, 0, st>>>(n, nr, bnr, bnc, pad, ptr, idx, count, count + nr); LAUNCH_CHECK(); }
    const int rc = exclusive_scan(nr, count, bptr, scratch, st);
    if (rc) return rc;
    int over = 0;
    HIP_TRY(hipMemcpyAsync(bnnz, bptr + nr, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&over, count + nr, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (over) *bnnz = -1;
    return 0;
}
extern "C" int liship_csr_to_bsr(int n, int bnr, int bnc, int bnnz, const int *ptr, const int *idx, const double *val, const int *bptr,
                                 int *bindex, double *bval, void *stream)
{
    hipStream_t st = as_stream(stream);
    const int nr = n > 0 ? 1 + (n - 1) / bnr : 0, pad = (bnc - n % bnc) % bnc;
    if (bnnz > 0) {
        HIP_TRY(hipMemsetAsync(bval, 0, sizeof(double) * (size_t)bnnz * (size_t)bnr * (size_t)bnc, st));
        HIP_TRY(hipMemsetAsync(bindex, 0xff, sizeof(int) * (size_t)bnnz, st));         // -1: no block column yet
        bsr_fill<<<grid_for(nr), BLOCK, 0, st>>>(n, nr, bnr, bnc, pad, ptr, idx, val, bptr, bindex, bval);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int liship_csr_to_jad(int n, const int *perm, const int *jptr, const int *ptr, const int *idx, const double *val,
                                 int *jidx, double *jval, void *stream)
{
    if (n > 0) { csr_to_jad<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, perm, jptr, ptr, idx, val, jidx, jval); LAUNCH_CHECK(); }
    return 0;
}
