// spmv_formats.hip -- ELL / DIA / JAD / BSR sparse matrix-vector products for gfx950, f64 / i32.
//
// One lane owns one output row and adds its terms strictly in the stored order starting from +0.0,
// which is the rounding sequence of the reference's loops (each `y[i] += a*x` there is one rounded
// multiply and one rounded add), so results are bit-identical to the reference at any thread count:
//   ELL  src/matvec/lis_matvec_ell.c:113-128      DIA  src/matvec/lis_matvec_dia.c:148-172
//   JAD  src/matvec/lis_matvec_jad.c:170-196      BSR  src/matvec/lis_matvec_bsr.c:120-148 (+ RxC :152-858)
// Column-major ELL/DIA/JAD storage makes the per-lane streams perfectly coalesced across a wavefront.
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK)
void spmv_ell_kernel(int n, int maxnzr, const int *__restrict__ idx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0;
    int j = 0;
    for (; j + 4 <= maxnzr; j += 4) {            // four independent (value,index,x) chains in flight
        double v[4]; int c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t k = (size_t)(j + u) * (size_t)n + (size_t)r;
            v[u] = load_stream(val + k);
            c[u] = load_stream(idx + k);
        }
        double xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u] * xv[u];
    }
    for (; j < maxnzr; j++) {
        const size_t k = (size_t)j * (size_t)n + (size_t)r;
        acc += load_stream(val + k) * x[load_stream(idx + k)];
    }
    y[r] = acc;
}

__global__ __launch_bounds__(BLOCK)
void spmv_dia_kernel(int n, int ncols, int nnd, const int *__restrict__ off,
                     const double *__restrict__ val, const double *__restrict__ x,
                     double *__restrict__ y)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0;
    for (int d = 0; d < nnd; d++) {
        const int c = r + off[d];                 // off[d] is wave-uniform -> scalar load
        if (c >= 0 && c < ncols)
            acc += load_stream(val + (size_t)d * (size_t)n + (size_t)r) * x[c];
    }
    y[r] = acc;
}

__global__ __launch_bounds__(BLOCK)
void spmv_jad_kernel(int n, int maxnzr, const int *__restrict__ perm, const int *__restrict__ ptr,
                     const int *__restrict__ idx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y)
{
    const int s = blockIdx.x * BLOCK + threadIdx.x;   // slot in the length-sorted order
    if (s >= n) return;
    double acc = 0.0;
    for (int j = 0; j < maxnzr; j++) {
        const int b = ptr[j], len = ptr[j + 1] - b;   // wave-uniform
        if (s >= len) break;                          // jagged diagonals only get shorter
        const int k = b + s;
        acc += load_stream(val + k) * x[load_stream(idx + k)];
    }
    y[perm[s]] = acc;
}

// one lane per scalar row: blocks of the block row in stored order, block columns ascending --
// the accumulation order of both the generic and the unrolled RxC reference kernels
__global__ __launch_bounds__(BLOCK)
void spmv_bsr_kernel(int nrows, int bnr, int bnc, const int *__restrict__ bptr,
                     const int *__restrict__ bidx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= nrows) return;
    const int br = r / bnr, ii = r - br * bnr, bs = bnr * bnc;
    double acc = 0.0;
    for (int b = bptr[br]; b < bptr[br + 1]; b++) {
        const double *blk = val + (size_t)b * (size_t)bs + ii;
        const double *xb = x + (size_t)bidx[b] * (size_t)bnc;
        for (int j = 0; j < bnc; j++) acc += blk[(size_t)j * bnr] * xb[j];
    }
    y[r] = acc;
}

inline int grid_for(int n) { return (n + BLOCK - 1) / BLOCK; }

} // namespace

extern "C" int liship_spmv_ell_f64(int n, int maxnzr, const int *idx, const double *val,
                                   const double *x, double *y, void *stream)
{
    if (n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    spmv_ell_kernel<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, idx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_spmv_dia_f64(int n, int ncols, int nnd, const int *off, const double *val,
                                   const double *x, double *y, void *stream)
{
    if (n < 0 || nnd < 0 || ncols < n) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    spmv_dia_kernel<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, off, val, x, y);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_spmv_jad_f64(int n, int maxnzr, const int *perm, const int *ptr, const int *idx,
                                   const double *val, const double *x, double *y, void *stream)
{
    if (n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    spmv_jad_kernel<<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, perm, ptr, idx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_spmv_bsr_f64(int nr, int bnr, int bnc, const int *bptr, const int *bidx,
                                   const double *val, const double *x, double *y, void *stream)
{
    if (nr < 0 || bnr < 1 || bnc < 1) return LISHIP_ERR_ARG;
    if (nr == 0) return 0;
    const long long rows = (long long)nr * bnr;
    if (rows > 0x7fffffffLL) return LISHIP_ERR_ARG;
    spmv_bsr_kernel<<<grid_for((int)rows), BLOCK, 0, as_stream(stream)>>>((int)rows, bnr, bnc, bptr, bidx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}
