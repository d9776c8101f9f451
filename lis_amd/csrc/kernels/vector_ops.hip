// vector_ops.hip -- the BLAS-1 kernels the Krylov loops are made of, for gfx950 (f64).
//
// Element-wise kernels restate src/vector/lis_vector_opv.c of the reference one expression each, with
// one rounded multiply + one rounded add (no FMA contraction: compiled with -ffp-contract=off), so
// their results are bit-identical to the CPU loops.  Fused forms (two axpys in one pass, the BiCGSTAB
// direction update, update + norm + dot) apply the SAME expressions in the SAME order per element --
// fusing removes passes over HBM, not roundings.
//
// Launch shape (measured, tools/ubench_axpy.hip, 1 GiB vectors): one-shot grid, 4 independent 16 B
// accesses per lane per array issued before the first use, non-temporal stores always, non-temporal
// loads once the vectors are larger than the 256 MiB Infinity Cache: 4.8 -> 6.1 TB/s for axpy.
//
// Reductions (src/vector/lis_vector_ops.c) are trees: every lane accumulates its slice, a wavefront
// butterfly (__shfl_xor) + LDS folds the workgroup, the per-workgroup partials are folded by further
// one-shot passes until one workgroup finishes.  The tree depends only on n, so a result is
// reproducible run to run; it is NOT the reference's left-to-right order (which itself changes with
// OMP_NUM_THREADS, SURVEY 7 "hard parts").
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;
constexpr int NW = BLOCK / WAVE;
constexpr int U = 4;                               // 16 B accesses per lane per array
constexpr int PAIRS_PER_BLOCK = BLOCK * U;         // 2048 doubles per workgroup
constexpr long long NT_LOAD_ELEMS = 32LL << 20;    // vectors beyond 256 MiB: stream past the caches
constexpr size_t MAX_PARTIALS = (1u << 20) + 4096; // level-1 partials of a 2^31-element vector (+ level 2)

inline int blocks_for(long long npairs) { return (int)((npairs + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK); }

template <bool NT> __device__ __forceinline__ v2f64 ld2(const double *p, long long pair)
{
    const v2f64 *q = reinterpret_cast<const v2f64 *>(p) + pair;
    return NT ? __builtin_nontemporal_load(q) : *q;
}
__device__ __forceinline__ void st2(double *p, long long pair, v2f64 v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<v2f64 *>(p) + pair);
}

// ---- device-driven Krylov loops ----------------------------------------------------------------------
// A solver that keeps its scalars in HBM (liship.h "krylov state") enqueues many iterations without a host
// round trip.  Two things make that safe: (1) coefficients can be read from HBM at kernel start (Dev::pa/pb),
// (2) every kernel launched while a guard is installed returns at once when *guard != 0 -- the state's DONE
// flag, raised by the scalar step that detects convergence -- so the iterations queued behind the converged
// one change nothing.
struct Dev { const double *pa, *pb, *skip; };
const double *g_guard = nullptr;

// ---- element-wise ---------------------------------------------------------------------------------
enum EwOp { EW_AXPY, EW_XPAY, EW_AXPYZ, EW_SCALE_TO, EW_PMUL, EW_PDIV, EW_SET, EW_ABS, EW_RECIP, EW_SHIFT,
            EW_AXPY2, EW_PUPDATE, EW_PMUL_XPAY, EW_SCALE_DEV, EW_RSQRT_ABS };

// x = in0, y = in1, w = in2; the expressions are the reference's (file:line in liship.h)
template <int OP>
__device__ __forceinline__ double ew_apply(double a, double b, double x, double y, double w)
{
    switch (OP) {
    case EW_AXPY:     return y + a * x;              // y[i] += alpha * x[i]
    case EW_XPAY:     return x + a * y;              // y[i] = x[i] + alpha * y[i]
    case EW_AXPYZ:    return a * x + y;              // z[i] = alpha * x[i] + y[i]
    case EW_SCALE_TO: return a * x;
    case EW_PMUL:     return x * y;
    case EW_PDIV:     return x / y;
    case EW_SET:      return a;
    case EW_ABS:      return fabs(x);
    case EW_RECIP:    return 1.0 / x;
    case EW_SHIFT:    return x - a;
    case EW_AXPY2:    { double t = y + a * x; return t + b * w; }        // y += a*x ; y += b*w
    case EW_PUPDATE:  { double t = y + a * x; return w + b * t; }        // y += a*x ; y = w + b*y
    case EW_PMUL_XPAY: { double z = x * w; return z + a * y; }          // z = x.*w ; y = z + a*y
    case EW_SCALE_DEV: return a * x;             // a = 1/sqrt(*device scalar), formed once per lane
    case EW_RSQRT_ABS: return 1.0 / sqrt(fabs(x));   // d[i] = 1.0 / sqrt(fabs(d[i]))  (lis_matrix_ops.c:611-614)
    }
    return 0.0;
}

template <int OP> struct EwArity { static constexpr int value =
    (OP == EW_SET) ? 0 : (OP == EW_SCALE_TO || OP == EW_ABS || OP == EW_RECIP || OP == EW_SHIFT || OP == EW_SCALE_DEV || OP == EW_RSQRT_ABS) ? 1 :
    (OP == EW_AXPY2 || OP == EW_PUPDATE || OP == EW_PMUL_XPAY) ? 3 : 2; };

template <int OP, bool NT, bool VEC>
__global__ __launch_bounds__(BLOCK)
void ew_kernel(int n, double a, double b, const double *in0, const double *in1, const double *in2, double *out, Dev D)
{
    constexpr int NIN = EwArity<OP>::value;
    if (D.skip && D.skip[0] != 0.0) return;
    if (D.pa) a = D.pa[0];
    if (D.pb) b = D.pb[0];
    if (OP == EW_SCALE_DEV) a = 1.0 / sqrt(in1[0]);     // in1: device scalar (a sum of squares), lis_solver_gmres.c:229-232
    if (VEC) {
        const long long npairs = n >> 1;
        const long long base = (long long)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x;
        v2f64 x[U], y[U], w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long p = base + u * BLOCK;
            if (p < npairs) {
                if (NIN >= 1) x[u] = ld2<NT>(in0, p);
                if (NIN >= 2) y[u] = ld2<NT>(in1, p);
                if (NIN >= 3) w[u] = ld2<NT>(in2, p);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long p = base + u * BLOCK;
            if (p < npairs) {
                v2f64 r;
                r.x = ew_apply<OP>(a, b, NIN >= 1 ? x[u].x : 0.0, NIN >= 2 ? y[u].x : 0.0, NIN >= 3 ? w[u].x : 0.0);
                r.y = ew_apply<OP>(a, b, NIN >= 1 ? x[u].y : 0.0, NIN >= 2 ? y[u].y : 0.0, NIN >= 3 ? w[u].y : 0.0);
                st2(out, p, r);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int i = n - 1;
            out[i] = ew_apply<OP>(a, b, NIN >= 1 ? in0[i] : 0.0, NIN >= 2 ? in1[i] : 0.0, NIN >= 3 ? in2[i] : 0.0);
        }
    } else {                                     // arrays that are only 8 B aligned
        const long long i0 = (long long)blockIdx.x * (2 * PAIRS_PER_BLOCK) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            const long long i = i0 + u * BLOCK;
            if (i < n) out[i] = ew_apply<OP>(a, b, NIN >= 1 ? in0[i] : 0.0, NIN >= 2 ? in1[i] : 0.0, NIN >= 3 ? in2[i] : 0.0);
        }
    }
}

template <int OP>
int run_ew(int n, double a, double b, const double *in0, const double *in1, const double *in2, double *out, void *stream,
           const double *pa = nullptr, const double *pb = nullptr)
{
    if (n < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    constexpr int NIN = EwArity<OP>::value;
    const bool vec = aligned16(out) && (NIN < 1 || aligned16(in0)) && (NIN < 2 || aligned16(in1)) && (NIN < 3 || aligned16(in2));
    const int grid = blocks_for(vec ? (n + 1) / 2 : ((long long)n + 1) / 2);
    hipStream_t st = as_stream(stream);
    const Dev D{pa, pb, g_guard};
    if (!vec)                       ew_kernel<OP, false, false><<<grid, BLOCK, 0, st>>>(n, a, b, in0, in1, in2, out, D);
    else if (n > NT_LOAD_ELEMS)     ew_kernel<OP, true, true><<<grid, BLOCK, 0, st>>>(n, a, b, in0, in1, in2, out, D);
    else                            ew_kernel<OP, false, true><<<grid, BLOCK, 0, st>>>(n, a, b, in0, in1, in2, out, D);
    LAUNCH_CHECK();
    return 0;
}

// CG's direction update with the PREVIOUS iteration's x update folded in (device-driven loop): x += alpha*p on the old p, then
// p = M^-1 r + beta*p -- lis_solver_cg.c:199 (axpy) and :176-183 (psolve none / Jacobi, xpay), each with its own rounding
// sequence.  p is read once for both (the pass that updates x in the reference's position reads it a second time).
// JAC: 0 no preconditioner, 1 Jacobi (z = r.*dinv), 2 Jacobi with a UNIFORM diagonal (every dinv[i] is the same double dc, so
// z = r*dc is the same product and the array is not read)
template <bool NT, bool VEC, int JAC, bool XUP>
__global__ __launch_bounds__(BLOCK)
void cg_direction_kernel(int n, const double *__restrict__ palpha, const double *__restrict__ pbeta, const double *__restrict__ r,
                         const double *__restrict__ dinv, double dc, double *p, double *x, const double *skip)
{
    if (skip && skip[0] != 0.0) return;
    const double alpha = XUP ? palpha[0] : 0.0, beta = pbeta[0];
    auto one = [&](double rv, double dv, double pv, double xv, double &po, double &xo) {
        if (XUP) xo = xv + alpha * pv;                   // x[i] += alpha * p[i]
        const double z = JAC == 1 ? rv * dv : JAC == 2 ? rv * dc : rv;       // z = M^-1 r
        po = z + beta * pv;                              // p[i] = z[i] + beta * p[i]
    };
    if (VEC) {
        const long long npairs = n >> 1;
        const long long base = (long long)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x;
        v2f64 rv[U], dv[U], pv[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long q = base + u * BLOCK;
            if (q < npairs) {
                rv[u] = ld2<NT>(r, q); pv[u] = ld2<NT>(p, q);
                if (JAC == 1) dv[u] = ld2<NT>(dinv, q);
                if (XUP) xv[u] = ld2<NT>(x, q);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long q = base + u * BLOCK;
            if (q < npairs) {
                double p0 = 0.0, p1 = 0.0, x0 = 0.0, x1 = 0.0;
                one(rv[u].x, JAC == 1 ? dv[u].x : 0.0, pv[u].x, XUP ? xv[u].x : 0.0, p0, x0);
                one(rv[u].y, JAC == 1 ? dv[u].y : 0.0, pv[u].y, XUP ? xv[u].y : 0.0, p1, x1);
                v2f64 po, xo;
                po.x = p0; po.y = p1; xo.x = x0; xo.y = x1;
                st2(p, q, po);
                if (XUP) st2(x, q, xo);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int i = n - 1;
            double po, xo;
            one(r[i], JAC == 1 ? dinv[i] : 0.0, p[i], XUP ? x[i] : 0.0, po, xo);
            p[i] = po;
            if (XUP) x[i] = xo;
        }
    } else {
        const long long i0 = (long long)blockIdx.x * (2 * PAIRS_PER_BLOCK) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            const long long i = i0 + u * BLOCK;
            if (i < n) {
                double po, xo;
                one(r[i], JAC == 1 ? dinv[i] : 0.0, p[i], XUP ? x[i] : 0.0, po, xo);
                p[i] = po;
                if (XUP) x[i] = xo;
            }
        }
    }
}

template <int JAC, bool XUP>
int run_cg_direction(int n, const double *palpha, const double *pbeta, const double *r, const double *dinv, double dc, double *p, double *x, void *stream)
{
    if (n < 0 || !pbeta || !r || !p || (XUP && (!palpha || !x)) || (JAC == 1 && !dinv)) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    const bool vec = aligned16(r) && aligned16(p) && (JAC != 1 || aligned16(dinv)) && (!XUP || aligned16(x));
    const int grid = blocks_for(((long long)n + 1) / 2);
    hipStream_t st = as_stream(stream);
    if (!vec)                   cg_direction_kernel<false, false, JAC, XUP><<<grid, BLOCK, 0, st>>>(n, palpha, pbeta, r, dinv, dc, p, x, g_guard);
    else if (n > NT_LOAD_ELEMS) cg_direction_kernel<true, true, JAC, XUP><<<grid, BLOCK, 0, st>>>(n, palpha, pbeta, r, dinv, dc, p, x, g_guard);
    else                        cg_direction_kernel<false, true, JAC, XUP><<<grid, BLOCK, 0, st>>>(n, palpha, pbeta, r, dinv, dc, p, x, g_guard);
    LAUNCH_CHECK();
    return 0;
}

// ---- reductions -------------------------------------------------------------------------------------
// level 1: partial[k*stride + block] for k < NRES; levels >= 2 sum partial arrays the same way
enum RedOp { RED_DOT, RED_SUMSQ, RED_ABS, RED_SUM, RED_DOT2,
             RED_CG_UPDATE,     // x += a*p; r += (-a)*q; result {sum r^2}
             RED_CG_UPDATE_JAC, // same + z = r.*dinv (not stored); results {sum r^2, sum r*z}
             RED_AXPY_NRM2,     // y += a*x; result {sum y^2}
             RED_AXPY_NRM2_DOT, // y += a*x; results {sum y^2, sum w*y}
             RED_AXPY_NRM2_JAC, // y += a*x; z = y.*e (not stored); results {sum y^2, sum y*z}
             RED_AXPY_NRM2_JACU,// the same with every e[i] equal to the double c (e is not read)
             RED_COUNT_NE,      // result {number of x[i] whose bits differ from a's}
             RED_BICGSTAB_END,  // e += (*pb)*d; e += (*pc)*y; y += a*x; results {sum y^2, sum w*y}   (x-iterate and residual of BiCGSTAB)
             RED_AXPYD_DOT,     // y += (-*sp)*x; result {sum y*w}      (one modified Gram-Schmidt step)
             RED_AXPYD_SUMSQ    // y += (-*sp)*x; result {sum y^2}      (the last one)
};
template <int OP> struct RedResults { static constexpr int value =
    (OP == RED_DOT2 || OP == RED_AXPY_NRM2_DOT || OP == RED_CG_UPDATE_JAC || OP == RED_AXPY_NRM2_JAC || OP == RED_AXPY_NRM2_JACU || OP == RED_BICGSTAB_END) ? 2 : 1; };

struct RedArgs {
    int n;
    double a;
    const double *x, *y, *w, *d, *e; // inputs (meaning per OP)
    double *ox, *oy;                 // in-place outputs of the fused forms
    const double *sp;                // device scalar of the *D ops (last: the other initialisers leave it NULL)
    const double *pa;                // coefficient `a` read from HBM when set (device-driven loops)
    const double *skip;              // guard flag (filled by run_reduce)
    const double *pb, *pc;           // further device coefficients (RED_BICGSTAB_END: alpha, omega)
    double c;                        // RED_AXPY_NRM2_JACU: the uniform 1/diag
};

// one element of a reduction: the fused forms' in-place results (ox, oy) and the term(s) this element adds to the sum(s).
// Shared by the tree kernels and the reference-order kernel, so the two orders add the SAME terms.
struct RedCoef { double a, adev, cb, cc, c; };
template <int OP>
__device__ __forceinline__ void red_term(const RedCoef &K, double x, double y, double w, double d, double e,
                                         double &ox, double &oy, double &v0, double &v1)
{
    v0 = 0.0; v1 = 0.0;
    if (OP == RED_DOT)   v0 = x * y;
    if (OP == RED_SUMSQ) v0 = x * x;
    if (OP == RED_ABS)   v0 = fabs(x);
    if (OP == RED_SUM)   v0 = x;
    if (OP == RED_DOT2) { v0 = x * y; v1 = x * x; }
    if (OP == RED_CG_UPDATE || OP == RED_CG_UPDATE_JAC) {   // x: p, y: q, w: x-iterate, d: residual r, e: 1/diag
        ox = w + K.a * x;               // x += alpha*p
        oy = d + (-K.a) * y;            // r += (-alpha)*q
        v0 = oy * oy;
        if (OP == RED_CG_UPDATE_JAC) { const double z = oy * e; v1 = oy * z; }   // z = r.*dinv ; <r,z>
    }
    if (OP == RED_COUNT_NE) v0 = (__double_as_longlong(x) != __double_as_longlong(K.a)) ? 1.0 : 0.0;
    if (OP == RED_AXPY_NRM2 || OP == RED_AXPY_NRM2_DOT || OP == RED_AXPY_NRM2_JAC || OP == RED_AXPY_NRM2_JACU) {
        oy = y + K.a * x;               // y += a*x
        v0 = oy * oy;
        if (OP == RED_AXPY_NRM2_DOT) v1 = w * oy;
        if (OP == RED_AXPY_NRM2_JAC) { const double z = oy * e; v1 = oy * z; }   // z = r.*dinv ; <r,z>
        if (OP == RED_AXPY_NRM2_JACU) { const double z = oy * K.c; v1 = oy * z; } // the same, dinv uniform
    }
    if (OP == RED_BICGSTAB_END) {       // x: t, y: s (becomes r), w: rtld, d: phat, e: the iterate; shat aliases s (no preconditioner)
        const double t1 = e + K.cb * d; // x += alpha*phat     (lis_solver_bicgstab.c:272)
        ox = t1 + K.cc * y;             // x += omega*shat     (:273)
        oy = y + K.a * x;               // r += (-omega)*t     (:276)
        v0 = oy * oy;
        v1 = w * oy;
    }
    if (OP == RED_AXPYD_DOT || OP == RED_AXPYD_SUMSQ) {
        oy = y + K.adev * x;            // y += (-h)*x, h read from HBM: no host round trip between steps
        if (OP == RED_AXPYD_DOT) v0 = oy * w; else v0 = oy * oy;
    }
}

template <int OP> struct RedShape {
    static constexpr bool IS_CG = (OP == RED_CG_UPDATE || OP == RED_CG_UPDATE_JAC || OP == RED_BICGSTAB_END);     // five inputs, two outputs
    static constexpr bool IS_AXD = (OP == RED_AXPYD_DOT || OP == RED_AXPYD_SUMSQ);
    static constexpr bool IS_AXN = (OP == RED_AXPY_NRM2 || OP == RED_AXPY_NRM2_DOT || OP == RED_AXPY_NRM2_JAC || OP == RED_AXPY_NRM2_JACU);
    static constexpr bool HAS_Y = (OP == RED_DOT || OP == RED_DOT2 || IS_CG || IS_AXN || IS_AXD);
    static constexpr bool HAS_W = (IS_CG || OP == RED_AXPY_NRM2_DOT || OP == RED_AXPYD_DOT);
    static constexpr bool HAS_D = IS_CG;
    static constexpr bool HAS_E = (OP == RED_CG_UPDATE_JAC || OP == RED_AXPY_NRM2_JAC || OP == RED_BICGSTAB_END);
};

template <int OP>
__device__ __forceinline__ RedCoef red_coef(RedArgs &A)
{
    if (A.pa) A.a = A.pa[0];
    RedCoef K;
    K.a = A.a;
    K.adev = (OP == RED_AXPYD_DOT || OP == RED_AXPYD_SUMSQ) ? -A.sp[0] : 0.0;
    K.cb = OP == RED_BICGSTAB_END ? A.pb[0] : 0.0;
    K.cc = OP == RED_BICGSTAB_END ? A.pc[0] : 0.0;
    K.c = A.c;
    return K;
}

template <int OP, bool NT, bool VEC>
__global__ __launch_bounds__(BLOCK)
void reduce_level1(RedArgs A, int stride, double *__restrict__ partial, bool root)
{
    __shared__ double scratch[NW];
    constexpr int NRES = RedResults<OP>::value;
    double s0 = 0.0, s1 = 0.0;
    const int n = A.n;
    if (A.skip && A.skip[0] != 0.0) return;
    const RedCoef K = red_coef<OP>(A);
    auto term = [&](double x, double y, double w, double d, double e, double &ox, double &oy) {
        double v0, v1;
        red_term<OP>(K, x, y, w, d, e, ox, oy, v0, v1);
        s0 += v0;
        if (NRES == 2) s1 += v1;
    };
    constexpr bool IS_CG = (OP == RED_CG_UPDATE || OP == RED_CG_UPDATE_JAC || OP == RED_BICGSTAB_END);     // five inputs, two outputs
    constexpr bool IS_AXD = (OP == RED_AXPYD_DOT || OP == RED_AXPYD_SUMSQ);
    constexpr bool IS_AXN = (OP == RED_AXPY_NRM2 || OP == RED_AXPY_NRM2_DOT || OP == RED_AXPY_NRM2_JAC || OP == RED_AXPY_NRM2_JACU);
    constexpr bool HAS_Y = (OP == RED_DOT || OP == RED_DOT2 || IS_CG || IS_AXN || IS_AXD);
    constexpr bool HAS_W = (IS_CG || OP == RED_AXPY_NRM2_DOT || OP == RED_AXPYD_DOT);
    constexpr bool HAS_D = IS_CG;
    constexpr bool HAS_E = (OP == RED_CG_UPDATE_JAC || OP == RED_AXPY_NRM2_JAC || OP == RED_BICGSTAB_END);
    if (VEC) {
        const long long npairs = n >> 1;
        const long long base = (long long)blockIdx.x * PAIRS_PER_BLOCK + threadIdx.x;
        v2f64 x[U], y[U], w[U], d[U], e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long p = base + u * BLOCK;
            if (p < npairs) {
                x[u] = ld2<NT>(A.x, p);
                if (HAS_Y) y[u] = ld2<NT>(A.y, p);
                if (HAS_W) w[u] = ld2<NT>(A.w, p);
                if (HAS_D) d[u] = ld2<NT>(A.d, p);
                if (HAS_E) e[u] = ld2<NT>(A.e, p);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long p = base + u * BLOCK;
            if (p < npairs) {
                double ox0 = 0.0, oy0 = 0.0, ox1 = 0.0, oy1 = 0.0;
                term(x[u].x, HAS_Y ? y[u].x : 0.0, HAS_W ? w[u].x : 0.0, HAS_D ? d[u].x : 0.0, HAS_E ? e[u].x : 0.0, ox0, oy0);
                term(x[u].y, HAS_Y ? y[u].y : 0.0, HAS_W ? w[u].y : 0.0, HAS_D ? d[u].y : 0.0, HAS_E ? e[u].y : 0.0, ox1, oy1);
                v2f64 ox, oy;
                ox.x = ox0; ox.y = ox1; oy.x = oy0; oy.y = oy1;
                if (IS_CG) { st2(A.ox, p, ox); st2(A.oy, p, oy); }
                if (IS_AXN || IS_AXD) st2(A.oy, p, oy);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int i = n - 1;
            double ox, oy;
            term(A.x[i], HAS_Y ? A.y[i] : 0.0, HAS_W ? A.w[i] : 0.0, HAS_D ? A.d[i] : 0.0, HAS_E ? A.e[i] : 0.0, ox, oy);
            if (IS_CG) { A.ox[i] = ox; A.oy[i] = oy; }
            if (IS_AXN || IS_AXD) A.oy[i] = oy;
        }
    } else {
        const long long i0 = (long long)blockIdx.x * (2 * PAIRS_PER_BLOCK) + threadIdx.x;
        for (int u = 0; u < 2 * U; u++) {
            const long long i = i0 + u * BLOCK;
            if (i < n) {
                double ox, oy;
                term(A.x[i], HAS_Y ? A.y[i] : 0.0, HAS_W ? A.w[i] : 0.0, HAS_D ? A.d[i] : 0.0, HAS_E ? A.e[i] : 0.0, ox, oy);
                if (IS_CG) { A.ox[i] = ox; A.oy[i] = oy; }
                if (IS_AXN || IS_AXD) A.oy[i] = oy;
            }
        }
    }
    const double t0 = block_sum<NW>(s0, scratch);       // root only when this block IS the whole reduction
    if (threadIdx.x == 0) partial[blockIdx.x] = root ? sqrt(t0) : t0;
    if (NRES == 2) {
        const double t1 = block_sum<NW>(s1, scratch);
        if (threadIdx.x == 0) partial[stride + blockIdx.x] = t1;
    }
}

// ---- reference-order reductions (liship_set_reference_reductions) ---------------------------------------
// The reference sums as its OpenMP build does (src/vector/lis_vector_ops.c:88-107 dot, :241-259 nrm2, nrm1 alike): thread t of T
// owns the contiguous chunk LIS_GET_ISIE(t, T, n) (include/lis.h:1067-1078 -- the static schedule of `omp for`), adds its terms
// strictly left to right from 0.0, and the T partial sums are added serially, thread 0 first, from 0.0.  Here one workgroup is one
// such thread: its 256 lanes form the terms of a tile (and store the fused forms' element-wise results), lane 0 adds the tile's
// terms in index order out of LDS (lane 64 the second result's), and reduce_ref_final adds the T partials in chunk order.  The
// result carries the reference's bits for OMP_NUM_THREADS = T; a parity mode, not a fast one (one lane adds at ~8 cycles a term).
// A multi-rank job whose ranks hold the row blocks LIS_GET_ISIE(rank, P, gn) at T = 1 forms, with the rank-order fold, the sum
// of a single rank at T = P.
constexpr int REF_TILE = 2048;
int g_ref_chunks = 0;

template <int OP>
__global__ __launch_bounds__(BLOCK)
void reduce_ref_kernel(RedArgs A, int T, double *__restrict__ partial)
{
    constexpr int NRES = RedResults<OP>::value;
    using S = RedShape<OP>;
    __shared__ double t0[REF_TILE];
    __shared__ double t1[NRES == 2 ? REF_TILE : 1];
    if (A.skip && A.skip[0] != 0.0) return;
    const RedCoef K = red_coef<OP>(A);
    const int n = A.n, c = blockIdx.x;
    long long is, ie;                                   // LIS_GET_ISIE(c, T, n, is, ie)
    if (c < n % T) { ie = n / T + 1; is = ie * c; } else { ie = n / T; is = ie * c + n % T; }
    ie += is;
    double s = 0.0;                                     // lane 0: the first result's running sum; lane 64: the second's
    for (long long base = is; base < ie; base += REF_TILE) {
        const int cnt = (int)(ie - base < REF_TILE ? ie - base : REF_TILE);
        for (int j = threadIdx.x; j < cnt; j += BLOCK) {
            const long long i = base + j;
            double ox = 0.0, oy = 0.0, v0, v1;
            red_term<OP>(K, A.x[i], S::HAS_Y ? A.y[i] : 0.0, S::HAS_W ? A.w[i] : 0.0, S::HAS_D ? A.d[i] : 0.0, S::HAS_E ? A.e[i] : 0.0,
                         ox, oy, v0, v1);
            if (S::IS_CG) { A.ox[i] = ox; A.oy[i] = oy; }
            if (S::IS_AXN || S::IS_AXD) A.oy[i] = oy;
            t0[j] = v0;
            if (NRES == 2) t1[j] = v1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll 8
            for (int j = 0; j < cnt; j++) s += t0[j];
        }
        if (NRES == 2 && threadIdx.x == WAVE) {
#pragma unroll 8
            for (int j = 0; j < cnt; j++) s += t1[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[c] = s;
    if (NRES == 2 && threadIdx.x == WAVE) partial[T + c] = s;
}

// One lane: the scalar statements between the vector passes, on the state block in HBM.  Every operation is the
// IEEE double operation the host loop performs (no contraction; sqrt and / are correctly rounded on gfx950).
__device__ void krylov_step_device(int step, double *st, double *rhist, const double *gathered, int nranks)
{
    st[LISHIP_KS_NOT_HALF] = 1.0;                    // (re-)armed by every step; only BICGSTAB_HALF may lower it
    if (st[LISHIP_KS_DONE] != 0.0) return;
    if (gathered) {                                  // cross-rank fold, rank order (lis_vector_ops.c:119,263)
        int slot = LISHIP_KS_DOT0, count = 1;
        switch (step) {
        case LISHIP_STEP_CG_RESID:        slot = LISHIP_KS_SUM0; break;
        case LISHIP_STEP_CG_RESID_PRE:    slot = LISHIP_KS_SUM0; count = 2; break;
        case LISHIP_STEP_BICGSTAB_HALF:   slot = LISHIP_KS_SUM0; break;
        case LISHIP_STEP_BICGSTAB_OMEGA:  count = 2; break;
        case LISHIP_STEP_BICGSTAB_RESID:  slot = LISHIP_KS_SUM0; count = 2; break;
        case LISHIP_STEP_BICG_RESID:      slot = LISHIP_KS_SUM0; break;
        case LISHIP_STEP_BICG_RHO:        slot = LISHIP_KS_SUM0; count = 2; break;
        default: break;
        }
        for (int k = 0; k < count; k++) {
            double s = 0.0;
            for (int r = 0; r < nranks; r++) s += gathered[r * count + k];
            st[slot + k] = s;
        }
    }
    auto finish_iteration = [&](double nrm2) {
        const double iter = st[LISHIP_KS_ITER] + 1.0;
        st[LISHIP_KS_ITER] = iter;
        st[LISHIP_KS_NRM2] = nrm2;
        st[LISHIP_KS_NHIST] = iter;
        if (rhist) rhist[(long long)iter] = nrm2;
    };
    auto stop = [&](double status) { st[LISHIP_KS_STATUS] = status; st[LISHIP_KS_DONE] = 1.0; };
    switch (step) {
    case LISHIP_STEP_CG_ALPHA: {                      // lis_solver_cg.c:193-204
        const double dot_pq = st[LISHIP_KS_DOT0];
        if (dot_pq == 0.0) { st[LISHIP_KS_ITER] += 1.0; stop(2.0); return; }
        const double alpha = st[LISHIP_KS_RHO] / dot_pq;
        st[LISHIP_KS_ALPHA] = alpha; st[LISHIP_KS_NALPHA] = -alpha;
        return;
    }
    case LISHIP_STEP_CG_RESID:
    case LISHIP_STEP_CG_RESID_PRE: {                  // :207-215, then :180-184 of the next iteration
        const double nrm2 = sqrt(st[LISHIP_KS_SUM0]) * st[LISHIP_KS_BNRM];
        finish_iteration(nrm2);
        if (st[LISHIP_KS_TOL] >= nrm2) { stop(1.0); return; }
        const double rho_old = st[LISHIP_KS_RHO];
        const double rho = st[step == LISHIP_STEP_CG_RESID_PRE ? LISHIP_KS_SUM1 : LISHIP_KS_SUM0];
        st[LISHIP_KS_RHO_OLD] = rho_old; st[LISHIP_KS_RHO] = rho;
        st[LISHIP_KS_BETA] = rho / rho_old;
        return;
    }
    case LISHIP_STEP_BICGSTAB_ALPHA: {                // :190-196 (the test of rho precedes everything the iteration
        const double rho = st[LISHIP_KS_RHO];         // changes in x and r), :226-230
        if (rho == 0.0) { st[LISHIP_KS_ITER] += 1.0; stop(2.0); return; }
        const double alpha = rho / st[LISHIP_KS_DOT0];
        st[LISHIP_KS_ALPHA] = alpha; st[LISHIP_KS_NALPHA] = -alpha;
        return;
    }
    case LISHIP_STEP_BICGSTAB_HALF: {                 // :236-258
        const double nrm2 = sqrt(st[LISHIP_KS_SUM0]) * st[LISHIP_KS_BNRM];
        if (nrm2 <= st[LISHIP_KS_TOL]) { finish_iteration(nrm2); st[LISHIP_KS_NOT_HALF] = 0.0; stop(1.0); }
        else st[LISHIP_KS_NRM2] = nrm2;
        return;
    }
    case LISHIP_STEP_BICGSTAB_OMEGA: {                // :267-269
        const double omega = st[LISHIP_KS_DOT0] / st[LISHIP_KS_DOT1];
        st[LISHIP_KS_OMEGA] = omega; st[LISHIP_KS_NOMEGA] = -omega;
        return;
    }
    case LISHIP_STEP_BICGSTAB_RESID: {                // :279-303
        const double nrm2 = sqrt(st[LISHIP_KS_SUM0]) * st[LISHIP_KS_BNRM];
        finish_iteration(nrm2);
        if (st[LISHIP_KS_TOL] >= nrm2) { stop(1.0); return; }
        if (st[LISHIP_KS_OMEGA] == 0.0) { stop(2.0); return; }
        const double rho_old = st[LISHIP_KS_RHO], rho = st[LISHIP_KS_SUM1];
        st[LISHIP_KS_RHO_OLD] = rho_old; st[LISHIP_KS_RHO] = rho;
        st[LISHIP_KS_BETA] = (rho / rho_old) * (st[LISHIP_KS_ALPHA] / st[LISHIP_KS_OMEGA]);   // :212 of the next iteration
        return;
    }
    case LISHIP_STEP_BICG_ALPHA: {                    // lis_solver_bicg.c:187-195, :226-238
        const double rho = st[LISHIP_KS_RHO], d1 = st[LISHIP_KS_DOT0];
        if (rho == 0.0 || d1 == 0.0) { st[LISHIP_KS_ITER] += 1.0; stop(2.0); return; }
        const double alpha = rho / d1;
        st[LISHIP_KS_ALPHA] = alpha; st[LISHIP_KS_NALPHA] = -alpha;
        return;
    }
    case LISHIP_STEP_BICG_RESID: {                    // :244-256
        const double nrm2 = sqrt(st[LISHIP_KS_SUM0]) * st[LISHIP_KS_BNRM];
        finish_iteration(nrm2);
        if (st[LISHIP_KS_TOL] >= nrm2) stop(1.0);
        return;
    }
    case LISHIP_STEP_BICG_RHO: {                      // :258-260, then :180-197 of the next iteration
        const double rho_old = st[LISHIP_KS_RHO], rho = st[LISHIP_KS_SUM1];
        st[LISHIP_KS_RHO_OLD] = rho_old; st[LISHIP_KS_RHO] = rho;
        st[LISHIP_KS_BETA] = rho / rho_old;
        return;
    }
    default: return;
    }
}

__global__ void krylov_step_kernel(int step, double *st, double *rhist, const double *gathered, int nranks)
{ krylov_step_device(step, st, rhist, gathered, nranks); }

// a scalar step announced with liship_krylov_chain rides in the last kernel of the next reduction (single-rank
// jobs: nothing has to be gathered between the fold and the step), saving a launch per reduction
struct Chain { int step; double *st, *rh; };
Chain g_chain{0, nullptr, nullptr};
inline Chain take_chain() { Chain c = g_chain; g_chain.step = 0; return c; }

// levels >= 2: out[k*ostride + block] = sum of in[k*istride + block*2048 ...), finally one block
__global__ __launch_bounds__(BLOCK)
void reduce_fold(int count, int nres, int istride, int ostride, const double *__restrict__ in, double *__restrict__ out,
                 bool root)
{
    __shared__ double scratch[NW];
    for (int k = 0; k < nres; k++) {
        const double *src = in + (size_t)k * istride;
        double s = 0.0;
        const long long i0 = (long long)blockIdx.x * (2 * PAIRS_PER_BLOCK) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            const long long i = i0 + u * BLOCK;
            if (i < count) s += src[i];
        }
        const double t = block_sum<NW>(s, scratch);
        if (threadIdx.x == 0) out[(size_t)k * ostride + blockIdx.x] = root ? sqrt(t) : t;
    }
}

// last level in one launch: a single 1024-lane workgroup sums up to 2^14 partials per result (lane-strided, then the
// fixed wave/LDS tree), (16 per lane: beyond that a second level of 256-lane workgroups is faster)
__global__ __launch_bounds__(1024)
void reduce_final(int count, int nres, int istride, const double *__restrict__ in, double *result, bool root, Chain chain)
{
    __shared__ double scratch[1024 / WAVE];
    for (int k = 0; k < nres; k++) {
        const double *src = in + (size_t)k * istride;
        double s = 0.0;
        for (int i = threadIdx.x; i < count; i += 1024) s += src[i];
        const double t = block_sum<1024 / WAVE>(s, scratch);
        if (threadIdx.x == 0) result[k] = root ? sqrt(t) : t;
    }
    if (chain.step && threadIdx.x == 0) krylov_step_device(chain.step, chain.st, chain.rh, nullptr, 0);
}

__global__ void finish_kernel(int nres, int stride, bool root, const double *__restrict__ in, double *result, Chain chain)
{
    if (threadIdx.x == 0) {
        for (int k = 0; k < nres; k++) { const double s = in[(size_t)k * stride]; result[k] = root ? sqrt(s) : s; }
        if (chain.step) krylov_step_device(chain.step, chain.st, chain.rh, nullptr, 0);
    }
}

// the serial sum over the T chunk partials (lis_vector_ops.c:103-107), the root of nrm2, the announced scalar step
__global__ void reduce_ref_final(int T, int nres, const double *__restrict__ partial, double *result, bool root, const double *skip, Chain chain)
{
    if (threadIdx.x != 0) return;
    if (!(skip && skip[0] != 0.0)) {
        for (int k = 0; k < nres; k++) {
            double s = 0.0;
            for (int c = 0; c < T; c++) s += partial[(size_t)k * T + c];
            result[k] = root ? sqrt(s) : s;
        }
    }
    if (chain.step) krylov_step_device(chain.step, chain.st, chain.rh, nullptr, 0);
}

// fold `count` partials per result (layout partial[k*stride + i]) down to result[k]; scratch2 lives behind them
int fold_partials(int count, int nres, int stride, double *partial, double *spare, double *result, bool root, hipStream_t st)
{
    const double *cur = partial;
    int cstride = stride;
    double *bufs[2] = {spare, partial};
    int which = 0;
    while (count > 1) {
        if (count <= (1 << 14)) {                   // last level: one workgroup, straight into result[k]
            reduce_final<<<1, 1024, 0, st>>>(count, nres, cstride, cur, result, root, take_chain());
            LAUNCH_CHECK();
            return 0;
        }
        const int blocks = (count + 2 * PAIRS_PER_BLOCK - 1) / (2 * PAIRS_PER_BLOCK);
        if (blocks == 1) {                          // (not reached any more: counts this small take the branch above)
            reduce_fold<<<1, BLOCK, 0, st>>>(count, nres, cstride, 1, cur, result, root);
            LAUNCH_CHECK();
            return 0;
        }
        double *dst = bufs[which];
        reduce_fold<<<blocks, BLOCK, 0, st>>>(count, nres, cstride, blocks, cur, dst, false);
        LAUNCH_CHECK();
        cur = dst; cstride = blocks; count = blocks;
        which ^= 1;
    }
    finish_kernel<<<1, 64, 0, st>>>(nres, cstride, root, cur, result, take_chain());    // a single partial (SpMV epilogue of one block)
    LAUNCH_CHECK();
    return 0;
}

template <int OP>
int run_reduce(RedArgs A, double *result, void *work, bool root, void *stream)
{
    if (A.n < 0 || !result || !work) return LISHIP_ERR_ARG;
    constexpr int NRES = RedResults<OP>::value;
    const int n = A.n;
    bool vec = aligned16(A.x);
    if (A.y) vec = vec && aligned16(A.y);
    if (A.w) vec = vec && aligned16(A.w);
    if (A.d) vec = vec && aligned16(A.d);
    if (A.e) vec = vec && aligned16(A.e);
    if (A.ox) vec = vec && aligned16(A.ox);
    if (A.oy) vec = vec && aligned16(A.oy);
    int grid = blocks_for(vec ? (n + 1) / 2 : ((long long)n + 1) / 2);
    if (grid < 1) grid = 1;
    double *partial = static_cast<double *>(work);
    double *spare = partial + 2 * MAX_PARTIALS;              // second half of the scratch
    hipStream_t st = as_stream(stream);
    A.skip = g_guard;
    if (g_ref_chunks > 0) {                                  // the reference's order (parity mode)
        const int T = g_ref_chunks;
        reduce_ref_kernel<OP><<<T, BLOCK, 0, st>>>(A, T, partial);
        LAUNCH_CHECK();
        reduce_ref_final<<<1, WAVE, 0, st>>>(T, NRES, partial, result, root, A.skip, take_chain());
        LAUNCH_CHECK();
        return 0;
    }
    const bool single = (grid == 1);                         // one block: it writes result[k] itself
    double *dst = single ? result : partial;
    const bool r1 = single && root;
    if (!vec)                   reduce_level1<OP, false, false><<<grid, BLOCK, 0, st>>>(A, grid, dst, r1);
    else if (n > NT_LOAD_ELEMS) reduce_level1<OP, true, true><<<grid, BLOCK, 0, st>>>(A, grid, dst, r1);
    else                        reduce_level1<OP, false, true><<<grid, BLOCK, 0, st>>>(A, grid, dst, r1);
    LAUNCH_CHECK();
    if (single) {                                            // no fold to ride in: the announced step gets its own launch
        const Chain c = take_chain();
        if (c.step) { krylov_step_kernel<<<1, 1, 0, st>>>(c.step, c.st, c.rh, nullptr, 0); LAUNCH_CHECK(); }
        return 0;
    }
    return fold_partials(grid, NRES, grid, partial, spare, result, root, st);
}

__global__ __launch_bounds__(BLOCK)
void gather_kernel(int count, const int *__restrict__ index, const double *__restrict__ x,
                   double *__restrict__ out)
{
    const int stride = gridDim.x * BLOCK;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) out[i] = x[index[i]];
}

__global__ __launch_bounds__(BLOCK)
void scatter_add_kernel(int count, const int *__restrict__ index, const double *__restrict__ src,
                        double *__restrict__ y)
{
    const int stride = gridDim.x * BLOCK;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) y[index[i]] += src[i];   // indices unique
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

// ELL: the first slot j with index[j*n + r] == r (padding carries index r and value 0 behind the row's entries: a row without a diagonal entry gets that 0),
// lis_matrix_get_diagonal_ell, src/matrix/lis_matrix_ell.c.  Slot-major arrays: the lanes of a wavefront read consecutive addresses of every slot.
__global__ __launch_bounds__(BLOCK)
void ell_diagonal_kernel(int n, int maxnzr, const int *__restrict__ idx, const double *__restrict__ val, double *__restrict__ d)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    double v = 0.0;
    for (int j = 0; j < maxnzr; j++)
        if (idx[(size_t)j * n + r] == r) { v = val[(size_t)j * n + r]; break; }
    d[r] = v;
}

// DIA: the stored diagonal with offset 0 (explicit zeros included), else 0: lis_matrix_get_diagonal_dia, src/matrix/lis_matrix_dia.c (one chunk: value[d*n + i])
__global__ __launch_bounds__(BLOCK)
void dia_diagonal_kernel(int n, int nnd, const int *__restrict__ offs, const double *__restrict__ val, double *__restrict__ d)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    double v = 0.0;
    for (int j = 0; j < nnd; j++)
        if (offs[j] == 0) { v = val[(size_t)j * n + r]; break; }
    d[r] = v;
}

} // namespace

// scratch: two ping-pong partial areas, each big enough for two results of a 2^31-element vector
extern "C" size_t liship_reduce_work_bytes(void) { return sizeof(double) * 2 * MAX_PARTIALS * 2; }

// T > 0: every reduction of this file is formed in the reference's order for OMP_NUM_THREADS = T (reduce_ref_kernel); 0: the trees.
// The products' fused-dot epilogues refuse while it is on (their callers then run the product and one reduction pass).
extern "C" int liship_set_reference_reductions(int T)
{
    if (T < 0 || (size_t)T > MAX_PARTIALS) return LISHIP_ERR_ARG;
    g_ref_chunks = T;
    return 0;
}
extern "C" int liship_get_reference_reductions(void) { return g_ref_chunks; }
int liship_internal_ref_chunks(void) { return g_ref_chunks; }

// used by spmv_csr.hip's fused dot epilogue
int liship_internal_fold(int count, int nres, int stride, double *partial, double *spare, double *result, void *stream)
{ return fold_partials(count, nres, stride, partial, spare, result, false, as_stream(stream)); }

extern "C" int liship_axpy_f64(int n, double a, const double *x, double *y, void *s)
{ return run_ew<EW_AXPY>(n, a, 0.0, x, y, nullptr, y, s); }
extern "C" int liship_xpay_f64(int n, const double *x, double a, double *y, void *s)
{ return run_ew<EW_XPAY>(n, a, 0.0, x, y, nullptr, y, s); }
extern "C" int liship_axpyz_f64(int n, double a, const double *x, const double *y, double *z, void *s)
{ return run_ew<EW_AXPYZ>(n, a, 0.0, x, y, nullptr, z, s); }
extern "C" int liship_scale_f64(int n, double a, double *x, void *s)
{ return run_ew<EW_SCALE_TO>(n, a, 0.0, x, nullptr, nullptr, x, s); }
extern "C" int liship_scale_to_f64(int n, double a, const double *x, double *y, void *s)
{ return run_ew<EW_SCALE_TO>(n, a, 0.0, x, nullptr, nullptr, y, s); }
extern "C" int liship_pmul_f64(int n, const double *x, const double *y, double *z, void *s)
{ return run_ew<EW_PMUL>(n, 0.0, 0.0, x, y, nullptr, z, s); }
extern "C" int liship_pdiv_f64(int n, const double *x, const double *y, double *z, void *s)
{ return run_ew<EW_PDIV>(n, 0.0, 0.0, x, y, nullptr, z, s); }
extern "C" int liship_set_all_f64(int n, double a, double *x, void *s)
{ return run_ew<EW_SET>(n, a, 0.0, nullptr, nullptr, nullptr, x, s); }
extern "C" int liship_abs_f64(int n, double *x, void *s)
{ return run_ew<EW_ABS>(n, 0.0, 0.0, x, nullptr, nullptr, x, s); }
extern "C" int liship_reciprocal_f64(int n, double *x, void *s)
{ return run_ew<EW_RECIP>(n, 0.0, 0.0, x, nullptr, nullptr, x, s); }
extern "C" int liship_rsqrt_abs_f64(int n, double *x, void *s)
{ return run_ew<EW_RSQRT_ABS>(n, 0.0, 0.0, x, nullptr, nullptr, x, s); }
extern "C" int liship_shift_f64(int n, double sigma, double *x, void *s)
{ return run_ew<EW_SHIFT>(n, sigma, 0.0, x, nullptr, nullptr, x, s); }
extern "C" int liship_axpy2_f64(int n, double a, const double *x, double b, const double *w, double *y, void *s)
{ return run_ew<EW_AXPY2>(n, a, b, x, y, w, y, s); }
extern "C" int liship_axpy_xpay_f64(int n, double a, const double *x, const double *w, double b, double *y, void *s)
{ return run_ew<EW_PUPDATE>(n, a, b, x, y, w, y, s); }

extern "C" int liship_dot_f64(int n, const double *x, const double *y, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, y, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_DOT>(A, r, w, false, s); }
extern "C" int liship_nrm2_f64(int n, const double *x, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_SUMSQ>(A, r, w, true, s); }
extern "C" int liship_sumsq_f64(int n, const double *x, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_SUMSQ>(A, r, w, false, s); }
extern "C" int liship_nrm1_f64(int n, const double *x, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_ABS>(A, r, w, false, s); }
extern "C" int liship_sum_f64(int n, const double *x, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_SUM>(A, r, w, false, s); }
extern "C" int liship_dot2_f64(int n, const double *x, const double *y, double *r, void *w, void *s)
{ RedArgs A{n, 0.0, x, y, nullptr, nullptr, nullptr, nullptr, nullptr}; return run_reduce<RED_DOT2>(A, r, w, false, s); }

// x += alpha*p ; r -= alpha*q ; result = {sum r^2}  (the two axpys + nrm2 of a CG iteration in one pass)
extern "C" int liship_cg_update_f64(int n, double alpha, const double *p, const double *q, double *x, double *r,
                                    double *result, void *w, void *s)
{
    RedArgs A{n, alpha, p, q, x, r, nullptr, x, r};
    return run_reduce<RED_CG_UPDATE>(A, result, w, false, s);
}
extern "C" int liship_cg_update_jacobi_f64(int n, double alpha, const double *p, const double *q, const double *dinv,
                                           double *x, double *r, double *result, void *w, void *s)
{
    RedArgs A{n, alpha, p, q, x, r, dinv, x, r};
    return run_reduce<RED_CG_UPDATE_JAC>(A, result, w, false, s);
}
extern "C" int liship_pmul_xpay_f64(int n, const double *x, const double *d, double a, double *y, void *s)
{ return run_ew<EW_PMUL_XPAY>(n, a, 0.0, x, y, d, y, s); }
// y += a*x ; result = {sum y^2}
extern "C" int liship_axpy_sumsq_f64(int n, double a, const double *x, double *y, double *result, void *w, void *s)
{ RedArgs A{n, a, x, y, nullptr, nullptr, nullptr, nullptr, y}; return run_reduce<RED_AXPY_NRM2>(A, result, w, false, s); }
// y += a*x ; result = {sum y^2, sum v*y}
extern "C" int liship_axpy_sumsq_dot_f64(int n, double a, const double *x, double *y, const double *v, double *result, void *w, void *s)
{ RedArgs A{n, a, x, y, v, nullptr, nullptr, nullptr, y}; return run_reduce<RED_AXPY_NRM2_DOT>(A, result, w, false, s); }

// modified Gram-Schmidt on the device: w += (-*hprev) * vprev, then result[0] = <w, vnext> (vnext != NULL) or
// sum w^2 (vnext == NULL).  hprev lives in HBM (the previous step's result), so a column of the Hessenberg is
// formed without a host synchronisation between steps.
extern "C" int liship_mgs_step_f64(int n, const double *hprev, const double *vprev, double *wv, const double *vnext,
                                   double *result, void *w, void *s)
{
    if (!hprev || !vprev) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, vprev, wv, vnext, nullptr, nullptr, nullptr, wv, hprev};
    if (vnext) return run_reduce<RED_AXPYD_DOT>(A, result, w, false, s);
    return run_reduce<RED_AXPYD_SUMSQ>(A, result, w, false, s);
}
// x *= 1/sqrt(*sumsq), sumsq in HBM  (lis_vector_nrm2 + lis_vector_scale, lis_solver_gmres.c:229-232)
extern "C" int liship_scale_inv_norm_f64(int n, const double *sumsq, double *x, void *s)
{ return run_ew<EW_SCALE_DEV>(n, 0.0, 0.0, x, sumsq, nullptr, x, s); }

// ---- device-driven Krylov loops: coefficient-from-HBM forms and the scalar steps (liship.h) --------------
extern "C" int liship_krylov_guard(const double *flag) { g_guard = flag; return 0; }
const double *liship_internal_guard(void) { return g_guard; }      // for the fused-product launches (spmv_*.hip)

extern "C" int liship_xpay_dev_f64(int n, const double *x, const double *pa, double *y, void *s)
{ return pa ? run_ew<EW_XPAY>(n, 0.0, 0.0, x, y, nullptr, y, s, pa) : LISHIP_ERR_ARG; }
extern "C" int liship_pmul_xpay_dev_f64(int n, const double *x, const double *d, const double *pa, double *y, void *s)
{ return pa ? run_ew<EW_PMUL_XPAY>(n, 0.0, 0.0, x, y, d, y, s, pa) : LISHIP_ERR_ARG; }
extern "C" int liship_axpy_dev_f64(int n, const double *pa, const double *x, double *y, void *s)
{ return pa ? run_ew<EW_AXPY>(n, 0.0, 0.0, x, y, nullptr, y, s, pa) : LISHIP_ERR_ARG; }
extern "C" int liship_axpy2_dev_f64(int n, const double *pa, const double *x, const double *pb, const double *w, double *y, void *s)
{ return (pa && pb) ? run_ew<EW_AXPY2>(n, 0.0, 0.0, x, y, w, y, s, pa, pb) : LISHIP_ERR_ARG; }
extern "C" int liship_axpy_xpay_dev_f64(int n, const double *pa, const double *x, const double *w, const double *pb, double *y, void *s)
{ return (pa && pb) ? run_ew<EW_PUPDATE>(n, 0.0, 0.0, x, y, w, y, s, pa, pb) : LISHIP_ERR_ARG; }
extern "C" int liship_cg_update_dev_f64(int n, const double *palpha, const double *p, const double *q, const double *dinv,
                                        double *x, double *r, double *result, void *w, void *s)
{
    if (!palpha) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, p, q, x, r, dinv, x, r, nullptr, palpha};
    return dinv ? run_reduce<RED_CG_UPDATE_JAC>(A, result, w, false, s) : run_reduce<RED_CG_UPDATE>(A, result, w, false, s);
}
// device-driven CG with the x update deferred into the next direction update (80 B instead of 88 B of vector traffic per row
// and iteration): palpha NULL = first iteration, x untouched; dinv NULL = no preconditioner
extern "C" int liship_cg_direction_dev_f64(int n, const double *palpha, const double *pbeta, const double *r, const double *dinv,
                                           double *p, double *x, void *s)
{
    if (dinv) return palpha ? run_cg_direction<1, true>(n, palpha, pbeta, r, dinv, 0.0, p, x, s) : run_cg_direction<1, false>(n, palpha, pbeta, r, dinv, 0.0, p, x, s);
    return palpha ? run_cg_direction<0, true>(n, palpha, pbeta, r, dinv, 0.0, p, x, s) : run_cg_direction<0, false>(n, palpha, pbeta, r, dinv, 0.0, p, x, s);
}
// the same for a Jacobi preconditioner whose diagonal is uniform: every dinv[i] is the double dc (the caller has checked: liship_count_ne_f64)
extern "C" int liship_cg_direction_uniform_dev_f64(int n, const double *palpha, const double *pbeta, const double *r, double dc,
                                                   double *p, double *x, void *s)
{
    return palpha ? run_cg_direction<2, true>(n, palpha, pbeta, r, nullptr, dc, p, x, s) : run_cg_direction<2, false>(n, palpha, pbeta, r, nullptr, dc, p, x, s);
}
// r += (*pna)*q ; result = {sum r^2, sum r*(r.*dinv)}   (the residual half of liship_cg_update_jacobi_f64; pna holds -alpha)
extern "C" int liship_cg_residual_jacobi_dev_f64(int n, const double *pna, const double *q, const double *dinv, double *r,
                                                 double *result, void *w, void *s)
{
    if (!pna || !dinv) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, q, r, nullptr, nullptr, dinv, nullptr, r, nullptr, pna};
    return run_reduce<RED_AXPY_NRM2_JAC>(A, result, w, false, s);
}
// the same when every dinv[i] is the double dc
extern "C" int liship_cg_residual_jacobi_uniform_dev_f64(int n, const double *pna, const double *q, double dc, double *r,
                                                         double *result, void *w, void *s)
{
    if (!pna) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, q, r, nullptr, nullptr, nullptr, nullptr, r, nullptr, pna};
    A.c = dc;
    return run_reduce<RED_AXPY_NRM2_JACU>(A, result, w, false, s);
}
// result[0] = number of x[i] that differ from `a` in any bit (0: the vector is uniform)
extern "C" int liship_count_ne_f64(int n, const double *x, double a, double *result, void *w, void *s)
{
    RedArgs A{n, a, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return run_reduce<RED_COUNT_NE>(A, result, w, false, s);
}
// BiCGSTAB without a preconditioner: x += (*palpha)*phat + (*pomega)*s ; r = s + (*pnomega)*t ; result = {sum r^2, sum rtld*r}
// (s is the residual array before the pass, r after it): liship_axpy2_dev_f64 + liship_axpy_sumsq_dot_dev_f64 in one pass
extern "C" int liship_bicgstab_end_dev_f64(int n, const double *palpha, const double *pomega, const double *pnomega, const double *phat,
                                           const double *t, const double *rtld, double *x, double *r, double *result, void *w, void *s)
{
    if (!palpha || !pomega || !pnomega) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, t, r, rtld, phat, x, x, r, nullptr, pnomega, nullptr, palpha, pomega};
    return run_reduce<RED_BICGSTAB_END>(A, result, w, false, s);
}
extern "C" int liship_axpy_sumsq_dev_f64(int n, const double *pa, const double *x, double *y, double *result, void *w, void *s)
{
    if (!pa) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, x, y, nullptr, nullptr, nullptr, nullptr, y, nullptr, pa};
    return run_reduce<RED_AXPY_NRM2>(A, result, w, false, s);
}
extern "C" int liship_axpy_sumsq_dot_dev_f64(int n, const double *pa, const double *x, double *y, const double *v,
                                             double *result, void *w, void *s)
{
    if (!pa) return LISHIP_ERR_ARG;
    RedArgs A{n, 0.0, x, y, v, nullptr, nullptr, nullptr, y, nullptr, pa};
    return run_reduce<RED_AXPY_NRM2_DOT>(A, result, w, false, s);
}

namespace {
__global__ void rank_fold_kernel(int count, const double *__restrict__ gathered, int nranks, double *__restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= count) return;
    double s = 0.0;
    for (int r = 0; r < nranks; r++) s += gathered[r * count + k];
    out[k] = s;
}
} // namespace
extern "C" int liship_rank_fold_f64(int count, const double *gathered, int nranks, double *out, void *stream)
{
    if (count < 1 || count > 64 || nranks < 1 || !gathered || !out) return LISHIP_ERR_ARG;
    rank_fold_kernel<<<1, 64, 0, as_stream(stream)>>>(count, gathered, nranks, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_krylov_chain(int step, double *state, double *rhistory)
{
    if (step == 0) { g_chain = Chain{0, nullptr, nullptr}; return 0; }      // withdraw an announced step (error unwinding)
    if (!state || step < LISHIP_STEP_CG_ALPHA || step > LISHIP_STEP_BICG_RHO) return LISHIP_ERR_ARG;
    g_chain = Chain{step, state, rhistory};
    return 0;
}
// the announced step did not find a reduction to ride in (a product without the fused epilogue, ...): run it alone
extern "C" int liship_krylov_chain_flush(void *stream)
{
    const Chain c = take_chain();
    if (c.step) { krylov_step_kernel<<<1, 1, 0, as_stream(stream)>>>(c.step, c.st, c.rh, nullptr, 0); LAUNCH_CHECK(); }
    return 0;
}
extern "C" int liship_krylov_step(int step, double *state, double *rhistory, const double *gathered, int nranks, void *stream)
{
    if (!state || step < LISHIP_STEP_CG_ALPHA || step > LISHIP_STEP_BICG_RHO) return LISHIP_ERR_ARG;
    krylov_step_kernel<<<1, 1, 0, as_stream(stream)>>>(step, state, rhistory, gathered, gathered ? nranks : 0);
    LAUNCH_CHECK();
    return 0;
}

namespace {
constexpr int LINCOMB_MAX = 48;
struct LinComb { int n, count, accumulate, aliased; const double *v[LINCOMB_MAX]; double c[LINCOMB_MAX]; };

// accumulate == 0: z = c0*v0; z += c1*v1; ...      (lis_vector_scale + axpys, lis_solver_gmres.c:290-296)
// accumulate == 1: z += c0*v0; z += c1*v1; ...     (a v may be z itself -- it then reads z as it was on entry:
//                 the residual update :323-329, the IDR(s) direction updates lis_solver_idrs.c:660-700)
__global__ __launch_bounds__(BLOCK)
void lincomb_kernel(LinComb L, double *__restrict__ z)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= L.n) return;
    const double z0 = (L.accumulate || L.aliased) ? z[i] : 0.0;    // a v that aliases z reads z as it was on entry
    double t;
    int j = 0;
    if (L.accumulate) t = z0;
    else { t = L.c[0] * ((L.v[0] == z) ? z0 : L.v[0][i]); j = 1; }
    for (; j < L.count; j++) {
        const double vj = (L.v[j] == z) ? z0 : L.v[j][i];
        t = t + L.c[j] * vj;
    }
    z[i] = t;
}
}

extern "C" int liship_lincomb_f64(int n, int count, const double *const *vs, const double *coef, int accumulate,
                                  double *z, void *s)
{
    if (n < 0 || count < 0 || !z) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    int done = 0;
    if (count == 0 && !accumulate) { HIP_TRY(hipMemsetAsync(z, 0, sizeof(double) * (size_t)n, as_stream(s))); return 0; }
    while (done < count) {
        LinComb L;
        L.n = n; L.accumulate = (done > 0) ? 1 : accumulate;
        L.count = (count - done < LINCOMB_MAX) ? count - done : LINCOMB_MAX;
        L.aliased = 0;
        for (int j = 0; j < L.count; j++) { L.v[j] = vs[done + j]; L.c[j] = coef[done + j]; if (L.v[j] == z) L.aliased = 1; }
        lincomb_kernel<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(s)>>>(L, z);
        LAUNCH_CHECK();
        done += L.count;
    }
    return 0;
}

extern "C" int liship_gather_f64(int count, const int *index, const double *x, double *out, void *s)
{
    if (count < 0) return LISHIP_ERR_ARG;
    if (count == 0) return 0;
    int grid = (count + BLOCK - 1) / BLOCK;
    if (grid > 4096) grid = 4096;
    gather_kernel<<<grid, BLOCK, 0, as_stream(s)>>>(count, index, x, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_scatter_add_f64(int count, const int *index, const double *src, double *y, void *s)
{
    if (count < 0) return LISHIP_ERR_ARG;
    if (count == 0) return 0;
    int grid = (count + BLOCK - 1) / BLOCK;
    if (grid > 4096) grid = 4096;
    scatter_add_kernel<<<grid, BLOCK, 0, as_stream(s)>>>(count, index, src, y);
    LAUNCH_CHECK();
    return 0;
}

namespace {
// value[j] *= d[i] (jacobi) or value[j] = value[j]*d[i]*d[index[j]] (symm_diag) for the entries of row i: lis_matrix_scale_csr /
// lis_matrix_scale_symm_csr (src/matrix/lis_matrix_csr.c:609-690), one rounded product after the other
__global__ __launch_bounds__(BLOCK)
void csr_scale_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ idx, double *__restrict__ val, const double *__restrict__ d, int symm)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const double dr = d[r];
    for (int k = ptr[r]; k < ptr[r + 1]; k++) {
        if (symm) { const double t = val[k] * dr; val[k] = t * d[idx[k]]; }
        else val[k] = val[k] * dr;
    }
}
} // namespace
extern "C" int liship_csr_scale_f64(int n, const int *ptr, const int *idx, double *val, const double *d, int symm, void *s)
{
    if (n < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    csr_scale_kernel<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(s)>>>(n, ptr, idx, val, d, symm);
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

extern "C" int liship_ell_diagonal_f64(int n, int maxnzr, const int *idx, const double *val, double *d, void *s)
{
    if (n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    ell_diagonal_kernel<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(s)>>>(n, maxnzr, idx, val, d);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_dia_diagonal_f64(int n, int nnd, const int *offs, const double *val, double *d, void *s)
{
    if (n < 0 || nnd < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    dia_diagonal_kernel<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, as_stream(s)>>>(n, nnd, offs, val, d);
    LAUNCH_CHECK();
    return 0;
}
