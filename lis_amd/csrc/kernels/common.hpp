// common.hpp -- shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LISHIP_ERR_ARG (-1)

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

// sum across the 64 lanes of a wavefront, fixed butterfly order (result valid in every lane)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
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
