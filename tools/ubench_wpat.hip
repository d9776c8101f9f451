// ubench_wpat.hip -- write-only patterns vs hipMemset (development tool): who owns which bytes matters for HBM writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2f64 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// PAT 0: grid-stride, one 16 B store per lane per step (wave = 1 KiB contiguous), any grid
// PAT 1: each workgroup owns ONE contiguous chunk of the buffer and streams through it (wave = 1 KiB per step)
// PAT 2: one-shot: workgroup writes U consecutive 4 KiB tiles (what the vector kernels do)
template <int PAT, bool NT, int U>
__global__ __launch_bounds__(256) void w(double *__restrict__ y, long long n2)
{
    v2f64 v; v.x = 1.0; v.y = 2.0;
    v2f64 *p = reinterpret_cast<v2f64 *>(y);
    if (PAT == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256)
            if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    } else if (PAT == 1) {
        const long long chunk = (n2 + gridDim.x - 1) / gridDim.x, b = chunk * blockIdx.x, e = b + chunk < n2 ? b + chunk : n2;
        for (long long i = b + threadIdx.x; i < e; i += 256)
            if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    } else {
        const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) { const long long i = base + u * 256; if (i < n2) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; } }
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
int main()
{
    const long long n = 1LL << 27, n2 = n / 2;       // 1 GiB
    double *y; CK(hipMalloc(&y, n * 8));
    printf("hipMemset: %.0f GB/s\n", n * 8 / timeit([&] { CK(hipMemsetAsync(y, 0, n * 8)); }) / 1e6);
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
        printf("grid %6d  stride plain %.0f nt %.0f | chunk plain %.0f nt %.0f GB/s\n", g,
               n * 8 / timeit([&] { w<0, false, 1><<<g, 256>>>(y, n2); }) / 1e6, n * 8 / timeit([&] { w<0, true, 1><<<g, 256>>>(y, n2); }) / 1e6,
               n * 8 / timeit([&] { w<1, false, 1><<<g, 256>>>(y, n2); }) / 1e6, n * 8 / timeit([&] { w<1, true, 1><<<g, 256>>>(y, n2); }) / 1e6);
    }
    printf("one-shot U=4: plain %.0f nt %.0f | U=8: plain %.0f nt %.0f | U=16 nt %.0f GB/s\n",
           n * 8 / timeit([&] { w<2, false, 4><<<(int)((n2 + 1023) / 1024), 256>>>(y, n2); }) / 1e6,
           n * 8 / timeit([&] { w<2, true, 4><<<(int)((n2 + 1023) / 1024), 256>>>(y, n2); }) / 1e6,
           n * 8 / timeit([&] { w<2, false, 8><<<(int)((n2 + 2047) / 2048), 256>>>(y, n2); }) / 1e6,
           n * 8 / timeit([&] { w<2, true, 8><<<(int)((n2 + 2047) / 2048), 256>>>(y, n2); }) / 1e6,
           n * 8 / timeit([&] { w<2, true, 16><<<(int)((n2 + 4095) / 4096), 256>>>(y, n2); }) / 1e6);
    return 0;
}
