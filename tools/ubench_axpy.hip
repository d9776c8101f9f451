// ubench_axpy.hip -- which store/load flavour and unroll suits the 2-read 1-write element-wise kernels (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2f64 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NTL, bool NTS, bool ONESHOT>
__global__ __launch_bounds__(256) void axpy(long long npairs, double a, const v2f64 *__restrict__ x, v2f64 *__restrict__ y)
{
    const long long stride = ONESHOT ? 0 : (long long)gridDim.x * 256 * U;
    for (long long base = (long long)blockIdx.x * 256 * U + threadIdx.x; base < npairs; base += stride) {
        v2f64 xv[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            long long p = base + u * 256;
            if (p < npairs) { xv[u] = NTL ? __builtin_nontemporal_load(x + p) : x[p]; yv[u] = NTL ? __builtin_nontemporal_load(y + p) : y[p]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            long long p = base + u * 256;
            if (p < npairs) {
                v2f64 r; r.x = yv[u].x + a * xv[u].x; r.y = yv[u].y + a * xv[u].y;
                if (NTS) __builtin_nontemporal_store(r, y + p); else y[p] = r;
            }
        }
        if (ONESHOT) break;
    }
}

template <typename F> float timeit(F f, int iters = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char **argv)
{
    long long n = (argc > 1 ? atoll(argv[1]) : 134217728LL);
    double *x, *y;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
    const long long np = n / 2;
#define RUN(U, NTL, NTS, ONE, GRID, name) { \
        float ms = timeit([&] { axpy<U, NTL, NTS, ONE><<<GRID, 256>>>(np, 0.5, (const v2f64 *)x, (v2f64 *)y); }); \
        printf("n=%lld %-40s grid=%8d %.4f ms %.0f GB/s\n", n, name, (int)(GRID), ms, 24.0 * n / ms / 1e6); }
    RUN(1, false, false, false, 4096, "U1 plain persistent4096");
    RUN(1, false, true, false, 4096, "U1 nt-store persistent4096");
    RUN(1, true, true, false, 4096, "U1 nt-load nt-store persistent4096");
    RUN(4, false, true, false, 2048, "U4 nt-store persistent2048");
    RUN(4, true, true, false, 2048, "U4 nt-load nt-store persistent2048");
    RUN(4, false, false, false, 2048, "U4 plain persistent2048");
    RUN(4, false, true, true, (int)((np + 1023) / 1024), "U4 nt-store one-shot");
    RUN(4, true, true, true, (int)((np + 1023) / 1024), "U4 nt-load nt-store one-shot");
    RUN(2, false, true, true, (int)((np + 511) / 512), "U2 nt-store one-shot");
    RUN(1, false, true, true, (int)((np + 255) / 256), "U1 nt-store one-shot");
    RUN(8, false, true, true, (int)((np + 2047) / 2048), "U8 nt-store one-shot");
    RUN(4, false, false, true, (int)((np + 1023) / 1024), "U4 plain one-shot");
    return 0;
}
