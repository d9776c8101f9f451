// ubench_stream.hip -- streaming ceilings on this box for the access mix of a CSR SpMV (development tool).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ub && /tmp/ub [nnz_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT, int UNROLL, bool WRITE_Y, bool IDX4>
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ val, const int *__restrict__ idx,
                                                 double *__restrict__ y, long long npairs, double *__restrict__ sink)
{
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    double acc = 0.0; int iacc = 0;
    for (long long base = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; base < npairs; base += stride) {
        v2f64 v[UNROLL]; v2i32 c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            long long p = base + (long long)u * 256; if (p >= npairs) p = npairs - 1;
            const v2f64 *pv = reinterpret_cast<const v2f64 *>(val) + p;
            const v2i32 *pc = reinterpret_cast<const v2i32 *>(idx) + p;
            v[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
            c[u] = NT ? __builtin_nontemporal_load(pc) : *pc;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { acc += v[u].x + v[u].y; iacc += c[u].x ^ c[u].y; }
        if (WRITE_Y) {   // 8 B per 7 nnz: 2*256*UNROLL nnz per iteration -> ~73*UNROLL rows
            long long row0 = (base - threadIdx.x) * 2 / 7;
            if (threadIdx.x < (2 * 256 * UNROLL) / 7) y[row0 + threadIdx.x] = acc;
        }
    }
    if (acc == 1.2345 && iacc == 77) sink[0] = acc;
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
    long long nnz = (argc > 1 ? atoll(argv[1]) : 938) * 1000000LL;
    nnz &= ~1LL;
    double *val, *y, *sink; int *idx;
    CK(hipMalloc(&val, nnz * 8)); CK(hipMalloc(&idx, nnz * 4)); CK(hipMalloc(&y, nnz / 7 * 8 + 65536)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(val, 0, nnz * 8)); CK(hipMemset(idx, 0, nnz * 4));
    const long long npairs = nnz / 2;
    const double gb = nnz * 12.0 / 1e9;
#define RUN(NT, U, W, GRID, name) { \
        float ms = timeit([&] { k_stream<NT, U, W, false><<<GRID, 256>>>(val, idx, y, npairs, sink); }); \
        double bytes = nnz * 12.0 + (W ? nnz / 7 * 8.0 : 0.0); \
        printf("%-44s grid=%8d  %.3f ms  %.0f GB/s\n", name, (int)(GRID), ms, bytes / ms / 1e6); }
    int oneshot4 = (int)((npairs + 256 * 4 - 1) / (256 * 4));
    RUN(true, 4, false, 2048, "nt  unroll4 read-only persistent2048");
    RUN(true, 4, false, 1024, "nt  unroll4 read-only persistent1024");
    RUN(true, 4, false, 4096, "nt  unroll4 read-only persistent4096");
    RUN(false, 4, false, 2048, "tmp unroll4 read-only persistent2048");
    RUN(true, 1, false, 2048, "nt  unroll1 read-only persistent2048");
    RUN(true, 2, false, 2048, "nt  unroll2 read-only persistent2048");
    RUN(true, 8, false, 2048, "nt  unroll8 read-only persistent2048");
    RUN(true, 8, false, 1024, "nt  unroll8 read-only persistent1024");
    RUN(true, 4, false, oneshot4, "nt  unroll4 read-only one-shot");
    RUN(true, 4, true, 2048, "nt  unroll4 +y-write persistent2048");
    RUN(false, 4, true, 2048, "tmp unroll4 +y-write persistent2048");
    RUN(true, 4, true, oneshot4, "nt  unroll4 +y-write one-shot");
    printf("total %.2f GB per pass\n", gb);
    return 0;
}
