// common.hpp -- shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LISHIP_ERR_ARG (-1)
#define LISHIP_ERR_TIMEOUT (-2)   // liship_stream_synchronize: the stream did not drain within the limit of liship_set_sync_timeout

// every launcher returns the hipError_t of the first failing runtime call
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

#define LAUNCH_CHECK()                                  \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

typedef double v2f64 __attribute__((ext_vector_type(2)));
// the same with the alignment of a double: two neighbouring x values gathered as one 16 B load from an address that is only 8 B aligned whenever
// the column is odd (the +-1 offsets of every stencil).  gfx9 global memory serves it as one dwordx4 in the default unaligned access mode;
// the type says so instead of leaving it to undefined behaviour.
typedef double v2f64u __attribute__((ext_vector_type(2), aligned(8)));
typedef int    v2i32 __attribute__((ext_vector_type(2)));
typedef int    v4i32 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;          // gfx950 wavefront
constexpr int NUM_XCD = 8;        // MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// streaming (read-once) loads: keep val[]/col[] from displacing x[] in L2 / Infinity Cache
template <typename T>
__device__ __forceinline__ T load_stream(const T *p) { return __builtin_nontemporal_load(p); }
// streaming (write-once) stores: measured on MI355X, a plain 8 B/row y store next to the value/index read
// streams costs 5.0 TB/s total where the nt form sustains 6.3 TB/s (tools/ubench_write.hip)
template <typename T>
__device__ __forceinline__ void store_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }

// the value of lane (i ^ off) in lane i, for off = 8, 4, 2, 1, by data-parallel-primitive moves inside a row of 16 lanes (no LDS
// permute): row_ror:8 IS i ^ 8; i ^ 4 is lane i + 4 for the banks of 4 lanes {0, 2} and lane i - 4 for the banks {1, 3};
// i ^ 2 and i ^ 1 are quad permutes.  All lanes must be active.
template <int OFF>
__device__ __forceinline__ double lane_xor_in_row(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (OFF == 8) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);          // row_ror:8
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    } else if (OFF == 4) {
        int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);      // row_shl:4 into banks 0 and 2: lane i + 4
        int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
        lo = __builtin_amdgcn_update_dpp(l2, lo, 0x114, 0xf, 0xa, false);         // row_shr:4 into banks 1 and 3: lane i - 4
        hi = __builtin_amdgcn_update_dpp(h2, hi, 0x114, 0xf, 0xa, false);
    } else if (OFF == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4e, 0xf, 0xf, false);           // quad_perm:[2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4e, 0xf, 0xf, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, false);           // quad_perm:[1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}

// sum across the 64 lanes of a wavefront, fixed butterfly order: partners 32, 16, 8, 4, 2, 1 lanes away, in that order (result
// valid in every lane).  The first two steps cross the rows of 16 lanes (LDS permute); the last four stay inside a row and use
// DPP moves of the SAME partner lanes -- the sum is the same in every bit (tests/golden/reduction_bits.json pins it).
__device__ __forceinline__ double wave_sum(double v)
{
    v += __shfl_xor(v, 32, WAVE);
    v += __shfl_xor(v, 16, WAVE);
    v += lane_xor_in_row<8>(v);
    v += lane_xor_in_row<4>(v);
    v += lane_xor_in_row<2>(v);
    v += lane_xor_in_row<1>(v);
    return v;
}

// block-wide sum for blockDim.x == 64*NW; scratch: NW doubles of LDS; result valid in thread 0
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; i++) t += scratch[i];
    }
    __syncthreads();
    return t;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// vector_ops.hip: fold `count` per-workgroup partials per result (layout partial[k*stride + i]) into result[k]
int liship_internal_fold(int count, int nres, int stride, double *partial, double *spare, double *result, void *stream);
// vector_ops.hip: the guard flag installed by liship_krylov_guard (NULL when none)
const double *liship_internal_guard(void);
// vector_ops.hip: T > 0 while the reductions are formed in the reference's order (liship_set_reference_reductions): the products' fused-dot epilogues refuse
int liship_internal_ref_chunks(void);
