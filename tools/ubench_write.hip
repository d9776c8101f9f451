// ubench_write.hip -- how expensive is the y write stream next to the value/index read streams? (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: no write; 1: plain 8B stores, 256-row aligned chunk per block-iteration; 2: nt stores; 3: unaligned (+3);
// 4: 16B stores by the first 128 lanes; 5: 16B nt stores; 6: rows chunk = 292 (all rows, as SpMV: 7 nnz/row) plain
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k(const double *__restrict__ val, const int *__restrict__ idx,
                                        double *__restrict__ y, long long npairs, double *__restrict__ sink)
{
    const long long nit = (npairs + 256 * UNROLL - 1) / (256 * UNROLL);
    double acc = 0.0; int iacc = 0;
    for (long long it = blockIdx.x; it < nit; it += gridDim.x) {
        const long long base = it * 256 * UNROLL + threadIdx.x;
        v2f64 v[UNROLL]; v2i32 c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            long long p = base + (long long)u * 256; if (p >= npairs) p = npairs - 1;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(val) + p);
            c[u] = __builtin_nontemporal_load(reinterpret_cast<const v2i32 *>(idx) + p);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { acc += v[u].x + v[u].y; iacc += c[u].x ^ c[u].y; }
        if (MODE == 1) y[it * 256 + threadIdx.x] = acc;
        if (MODE == 2) __builtin_nontemporal_store(acc, &y[it * 256 + threadIdx.x]);
        if (MODE == 3) y[it * 256 + threadIdx.x + 3] = acc;
        if (MODE == 4 && threadIdx.x < 128) { v2f64 w; w.x = acc; w.y = acc; reinterpret_cast<v2f64 *>(y)[it * 128 + threadIdx.x] = w; }
        if (MODE == 5 && threadIdx.x < 128) { v2f64 w; w.x = acc; w.y = acc; __builtin_nontemporal_store(w, &reinterpret_cast<v2f64 *>(y)[it * 128 + threadIdx.x]); }
        if (MODE == 6) { y[it * 292 + threadIdx.x] = acc; if (threadIdx.x < 36) y[it * 292 + 256 + threadIdx.x] = acc; }
        if (MODE == 7) { __builtin_nontemporal_store(acc, &y[it * 292 + threadIdx.x]); if (threadIdx.x < 36) __builtin_nontemporal_store(acc, &y[it * 292 + 256 + threadIdx.x]); }
    }
    if (acc == 1.2345 && iacc == 77) sink[0] = acc;
}

template <int W> __global__ __launch_bounds__(256) void kw(double *__restrict__ y, long long n2)
{
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n2; p += (long long)gridDim.x * 256) {
        v2f64 w; w.x = 1.0; w.y = 2.0;
        if (W == 0) reinterpret_cast<v2f64 *>(y)[p] = w; else __builtin_nontemporal_store(w, &reinterpret_cast<v2f64 *>(y)[p]);
    }
}

template <typename F> float timeit(F f, int iters = 10)
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
    long long nnz = (argc > 1 ? atoll(argv[1]) : 938) * 1000000LL; nnz &= ~1LL;
    double *val, *y, *sink; int *idx;
    const long long ny = nnz / 7 + (1 << 20);
    CK(hipMalloc(&val, nnz * 8)); CK(hipMalloc(&idx, nnz * 4)); CK(hipMalloc(&y, ny * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(val, 0, nnz * 8)); CK(hipMemset(idx, 0, nnz * 4));
    const long long npairs = nnz / 2;
#define RUN(MODE, U, GRID, name) { \
        float ms = timeit([&] { k<MODE, U><<<GRID, 256>>>(val, idx, y, npairs, sink); }); \
        long long nit = (npairs + 256 * U - 1) / (256 * U); \
        double wbytes = (MODE == 0) ? 0.0 : ((MODE >= 6) ? nit * 292 * 8.0 : nit * 256 * 8.0); \
        printf("%-40s grid=%7d  %.3f ms  read+write %.0f GB/s (write %.2f GB)\n", name, (int)(GRID), ms, (nnz * 12.0 + wbytes) / ms / 1e6, wbytes / 1e9); }
    long long nit4 = (npairs + 1023) / 1024;
    RUN(0, 4, 2048, "no write persistent");
    RUN(0, 4, (int)nit4, "no write one-shot");
    RUN(1, 4, 2048, "plain 8B aligned persistent");
    RUN(1, 4, (int)nit4, "plain 8B aligned one-shot");
    RUN(2, 4, 2048, "nt 8B aligned persistent");
    RUN(2, 4, (int)nit4, "nt 8B aligned one-shot");
    RUN(3, 4, (int)nit4, "plain 8B unaligned(+3) one-shot");
    RUN(4, 4, (int)nit4, "plain 16B one-shot");
    RUN(5, 4, (int)nit4, "nt 16B one-shot");
    RUN(6, 4, (int)nit4, "plain 8B 292 rows/iter one-shot");
    RUN(7, 4, (int)nit4, "nt 8B 292 rows/iter one-shot");
    RUN(6, 4, 2048, "plain 8B 292 rows/iter persistent");
    { float ms = timeit([&] { kw<0><<<4096, 256>>>(y, ny / 2); }); printf("write-only plain 16B: %.3f ms %.0f GB/s\n", ms, ny * 8.0 / ms / 1e6); }
    { float ms = timeit([&] { kw<1><<<4096, 256>>>(y, ny / 2); }); printf("write-only nt 16B:    %.3f ms %.0f GB/s\n", ms, ny * 8.0 / ms / 1e6); }
    { float ms = timeit([&] { CK(hipMemsetAsync(y, 0, ny * 8)); }); printf("hipMemset:            %.3f ms %.0f GB/s\n", ms, ny * 8.0 / ms / 1e6); }
    return 0;
}
