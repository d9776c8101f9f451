// vector_ops.hip -- the BLAS-1 kernels the Krylov loops are made of, for gfx950 (f64).
//
// Element-wise kernels restate src/vector/lis_vector_opv.c of the reference one expression each, with
// one rounded multiply + one rounded add (no FMA contraction: compiled with -ffp-contract=off), so
// their results are bit-identical to the CPU loops.  They stream 16 B per lane per access.
//
// Reductions (src/vector/lis_vector_ops.c) are two-stage: every lane accumulates a grid-strided
// slice, a wavefront butterfly (__shfl_xor) + LDS folds the workgroup, and a second one-workgroup
// kernel folds the REDUCE_GRID partials in a fixed order.  The grid is a constant, so a result is
// reproducible run to run and independent of the launch; it is NOT the reference's left-to-right
// order (which itself changes with OMP_NUM_THREADS, SURVEY 7 "hard parts").
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;
constexpr int NW = BLOCK / WAVE;
constexpr int EW_MAX_GRID = 256 * 16;        // element-wise: <= 16 workgroups per CU, grid-stride beyond
constexpr int REDUCE_GRID = 2048;            // partials per reduction (8 workgroups per CU)
constexpr int MAX_RESULTS = 2;               // dot2 produces two sums

inline int ew_grid(long long work_items)
{
    long long g = (work_items + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    if (g > EW_MAX_GRID) g = EW_MAX_GRID;
    return (int)g;
}

// ---- element-wise -------------------------------------------------------------------------
// F: functor called per element with (i, index) access to arrays hidden in the functor itself.
// Vector body works on aligned pairs; a scalar tail / unaligned fallback covers the rest.
template <typename F>
__global__ __launch_bounds__(BLOCK) void ew_kernel(int n, bool vec, F f)
{
    const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    if (vec) {
        const int npairs = n >> 1;
        for (int p = tid; p < npairs; p += stride) f.pair(p);
        if ((n & 1) && tid == 0) f.one(n - 1);
    } else {
        for (int i = tid; i < n; i += stride) f.one(i);
    }
}

#define V2(p) (reinterpret_cast<v2f64 *>(p))
#define CV2(p) (reinterpret_cast<const v2f64 *>(p))

struct AxpyF {   // y += a*x
    double a; const double *x; double *y;
    __device__ void one(int i) const { y[i] += a * x[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv = V2(y)[p]; yv.x += a * xv.x; yv.y += a * xv.y; V2(y)[p] = yv; }
};
struct XpayF {   // y = x + a*y
    double a; const double *x; double *y;
    __device__ void one(int i) const { y[i] = x[i] + a * y[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv = V2(y)[p]; yv.x = xv.x + a * yv.x; yv.y = xv.y + a * yv.y; V2(y)[p] = yv; }
};
struct AxpyzF {  // z = a*x + y
    double a; const double *x; const double *y; double *z;
    __device__ void one(int i) const { z[i] = a * x[i] + y[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv = CV2(y)[p], zv; zv.x = a * xv.x + yv.x; zv.y = a * xv.y + yv.y; V2(z)[p] = zv; }
};
struct ScaleToF { // y = a*x
    double a; const double *x; double *y;
    __device__ void one(int i) const { y[i] = a * x[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv; yv.x = a * xv.x; yv.y = a * xv.y; V2(y)[p] = yv; }
};
struct PmulF {   // z = x*y
    const double *x; const double *y; double *z;
    __device__ void one(int i) const { z[i] = x[i] * y[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv = CV2(y)[p], zv; zv.x = xv.x * yv.x; zv.y = xv.y * yv.y; V2(z)[p] = zv; }
};
struct PdivF {   // z = x/y
    const double *x; const double *y; double *z;
    __device__ void one(int i) const { z[i] = x[i] / y[i]; }
    __device__ void pair(int p) const { v2f64 xv = CV2(x)[p], yv = CV2(y)[p], zv; zv.x = xv.x / yv.x; zv.y = xv.y / yv.y; V2(z)[p] = zv; }
};
struct SetAllF {
    double a; double *x;
    __device__ void one(int i) const { x[i] = a; }
    __device__ void pair(int p) const { v2f64 v; v.x = a; v.y = a; V2(x)[p] = v; }
};
struct AbsF {
    double *x;
    __device__ void one(int i) const { x[i] = fabs(x[i]); }
    __device__ void pair(int p) const { v2f64 v = V2(x)[p]; v.x = fabs(v.x); v.y = fabs(v.y); V2(x)[p] = v; }
};
struct RecipF {
    double *x;
    __device__ void one(int i) const { x[i] = 1.0 / x[i]; }
    __device__ void pair(int p) const { v2f64 v = V2(x)[p]; v.x = 1.0 / v.x; v.y = 1.0 / v.y; V2(x)[p] = v; }
};
struct ShiftF {  // x = x - sigma
    double s; double *x;
    __device__ void one(int i) const { x[i] = x[i] - s; }
    __device__ void pair(int p) const { v2f64 v = V2(x)[p]; v.x = v.x - s; v.y = v.y - s; V2(x)[p] = v; }
};

template <typename F>
int run_ew(int n, bool vec, F f, void *stream)
{
    if (n < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    ew_kernel<F><<<ew_grid(vec ? (n + 1) / 2 : n), BLOCK, 0, as_stream(stream)>>>(n, vec, f);
    LAUNCH_CHECK();
    return 0;
}

// ---- reductions -----------------------------------------------------------------------------
enum RedOp { RED_DOT = 0, RED_SUMSQ = 1, RED_ABS = 2, RED_SUM = 3, RED_DOT2 = 4 };

template <int OP>
__device__ __forceinline__ void red_term(double xv, double yv, double &a0, double &a1)
{
    if (OP == RED_DOT)   a0 += xv * yv;
    if (OP == RED_SUMSQ) a0 += xv * xv;
    if (OP == RED_ABS)   a0 += fabs(xv);
    if (OP == RED_SUM)   a0 += xv;
    if (OP == RED_DOT2) { a0 += xv * yv; a1 += xv * xv; }
}

template <int OP>
__global__ __launch_bounds__(BLOCK)
void reduce_stage1(int n, bool vec, const double *__restrict__ x, const double *__restrict__ y,
                   double *__restrict__ partial)
{
    __shared__ double scratch[NW];
    constexpr bool TWO_IN = (OP == RED_DOT || OP == RED_DOT2);
    const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    double a0 = 0.0, a1 = 0.0;
    if (vec) {
        const int npairs = n >> 1;
        for (int p = tid; p < npairs; p += stride) {
            const v2f64 xv = CV2(x)[p];
            v2f64 yv = xv;
            if (TWO_IN) yv = CV2(y)[p];
            red_term<OP>(xv.x, yv.x, a0, a1);
            red_term<OP>(xv.y, yv.y, a0, a1);
        }
        if ((n & 1) && tid == 0) red_term<OP>(x[n - 1], TWO_IN ? y[n - 1] : 0.0, a0, a1);
    } else {
        for (int i = tid; i < n; i += stride) red_term<OP>(x[i], TWO_IN ? y[i] : 0.0, a0, a1);
    }
    const double s0 = block_sum<NW>(a0, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s0;
    if (OP == RED_DOT2) {
        const double s1 = block_sum<NW>(a1, scratch);
        if (threadIdx.x == 0) partial[REDUCE_GRID + blockIdx.x] = s1;
    }
}

// one workgroup: result[k] = fold(partial[k*REDUCE_GRID .. +count)), optional sqrt
__global__ __launch_bounds__(BLOCK)
void reduce_stage2(int count, int nresults, bool root, const double *__restrict__ partial,
                   double *__restrict__ result)
{
    __shared__ double scratch[NW];
    for (int k = 0; k < nresults; k++) {
        double a = 0.0;
        for (int i = threadIdx.x; i < count; i += BLOCK) a += partial[k * REDUCE_GRID + i];
        const double s = block_sum<NW>(a, scratch);
        if (threadIdx.x == 0) result[k] = root ? sqrt(s) : s;
    }
}

template <int OP>
int run_reduce(int n, const double *x, const double *y, double *result, void *work, bool root, void *stream)
{
    if (n < 0 || !result || !work) return LISHIP_ERR_ARG;
    constexpr bool TWO_IN = (OP == RED_DOT || OP == RED_DOT2);
    const bool vec = aligned16(x) && (!TWO_IN || aligned16(y));
    long long items = vec ? (n + 1) / 2 : n;
    int grid = (int)((items + BLOCK - 1) / BLOCK);
    if (grid < 1) grid = 1;
    if (grid > REDUCE_GRID) grid = REDUCE_GRID;
    double *partial = static_cast<double *>(work);
    hipStream_t st = as_stream(stream);
    reduce_stage1<OP><<<grid, BLOCK, 0, st>>>(n, vec, x, y, partial);
    LAUNCH_CHECK();
    reduce_stage2<<<1, BLOCK, 0, st>>>(grid, OP == RED_DOT2 ? 2 : 1, root, partial, result);
    LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(BLOCK)
void gather_kernel(int count, const int *__restrict__ index, const double *__restrict__ x,
                   double *__restrict__ out)
{
    const int stride = gridDim.x * BLOCK;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) out[i] = x[index[i]];
}

__global__ __launch_bounds__(BLOCK)
void csr_diagonal_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ idx,
                         const double *__restrict__ val, double *__restrict__ d)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    double v = 0.0;
    for (int k = ptr[r]; k < ptr[r + 1]; k++)
        if (idx[k] == r) { v = val[k]; break; }
    d[r] = v;
}

} // namespace

extern "C" size_t liship_reduce_work_bytes(void) { return sizeof(double) * REDUCE_GRID * MAX_RESULTS; }

extern "C" int liship_axpy_f64(int n, double a, const double *x, double *y, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y), AxpyF{a, x, y}, s); }
extern "C" int liship_xpay_f64(int n, const double *x, double a, double *y, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y), XpayF{a, x, y}, s); }
extern "C" int liship_axpyz_f64(int n, double a, const double *x, const double *y, double *z, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y) && aligned16(z), AxpyzF{a, x, y, z}, s); }
extern "C" int liship_scale_f64(int n, double a, double *x, void *s)
{ return run_ew(n, aligned16(x), ScaleToF{a, x, x}, s); }
extern "C" int liship_scale_to_f64(int n, double a, const double *x, double *y, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y), ScaleToF{a, x, y}, s); }
extern "C" int liship_pmul_f64(int n, const double *x, const double *y, double *z, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y) && aligned16(z), PmulF{x, y, z}, s); }
extern "C" int liship_pdiv_f64(int n, const double *x, const double *y, double *z, void *s)
{ return run_ew(n, aligned16(x) && aligned16(y) && aligned16(z), PdivF{x, y, z}, s); }
extern "C" int liship_set_all_f64(int n, double a, double *x, void *s)
{ return run_ew(n, aligned16(x), SetAllF{a, x}, s); }
extern "C" int liship_abs_f64(int n, double *x, void *s)
{ return run_ew(n, aligned16(x), AbsF{x}, s); }
extern "C" int liship_reciprocal_f64(int n, double *x, void *s)
{ return run_ew(n, aligned16(x), RecipF{x}, s); }
extern "C" int liship_shift_f64(int n, double sigma, double *x, void *s)
{ return run_ew(n, aligned16(x), ShiftF{sigma, x}, s); }

extern "C" int liship_dot_f64(int n, const double *x, const double *y, double *r, void *w, void *s)
{ return run_reduce<RED_DOT>(n, x, y, r, w, false, s); }
extern "C" int liship_nrm2_f64(int n, const double *x, double *r, void *w, void *s)
{ return run_reduce<RED_SUMSQ>(n, x, nullptr, r, w, true, s); }
extern "C" int liship_sumsq_f64(int n, const double *x, double *r, void *w, void *s)
{ return run_reduce<RED_SUMSQ>(n, x, nullptr, r, w, false, s); }
extern "C" int liship_nrm1_f64(int n, const double *x, double *r, void *w, void *s)
{ return run_reduce<RED_ABS>(n, x, nullptr, r, w, false, s); }
extern "C" int liship_sum_f64(int n, const double *x, double *r, void *w, void *s)
{ return run_reduce<RED_SUM>(n, x, nullptr, r, w, false, s); }
extern "C" int liship_dot2_f64(int n, const double *x, const double *y, double *r, void *w, void *s)
{ return run_reduce<RED_DOT2>(n, x, y, r, w, false, s); }

extern "C" int liship_gather_f64(int count, const int *index, const double *x, double *out, void *s)
{
    if (count < 0) return LISHIP_ERR_ARG;
    if (count == 0) return 0;
    gather_kernel<<<ew_grid(count), BLOCK, 0, as_stream(s)>>>(count, index, x, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_csr_diagonal_f64(int n, const int *ptr, const int *idx, const double *val,
                                       double *d, void *s)
{
    if (n < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    csr_diagonal_kernel<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(s)>>>(n, ptr, idx, val, d);
    LAUNCH_CHECK();
    return 0;
}
