// gather_policy_proto.hip -- experiment: what does a RANDOM 8 B gather cost at the L2 <-> fabric boundary, and does the cache policy of the load change the size of the
// request the L2 sends out?  (TCC_EA0_RDREQ_{32B,64B,128B} count them separately on gfx950.)  x: N doubles, idx: M random ints; every lane sums x[idx[k]] over its share.
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy_proto tools/proto/gather_policy_proto.hip && ./gather_policy_proto [N=2000000] [M=52000000]
// Under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum the kernels' names tell the policies apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

template <int POLICY> __device__ __forceinline__ double ld(const double *p)
{
    double v;
    if (POLICY == 0) return *p;
    if (POLICY == 1) return __builtin_nontemporal_load(p);
    if (POLICY == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 6) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 7) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 8) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// 8 independent gathers per lane and step (the inline-asm forms wait per load: they measure the POLICY's request size under the counters, their time is not comparable)
template <int POLICY>
__global__ __launch_bounds__(256) void gather(long long m, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, T = (long long)gridDim.x * 256;
    double s = 0.0;
    for (long long k = t; k + 7 * T < m; k += 8 * T) {
        int c[8]; double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = __builtin_nontemporal_load(idx + k + u * T);
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ld<POLICY>(x + c[u]);
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    out[t] = s;
}

template <int POLICY> static void run(const char *name, long long m, const int *idx, const double *x, double *out, hipEvent_t e0, hipEvent_t e1)
{
    const int grid = 256 * 32;
    for (int w = 0; w < 2; w++) gather<POLICY><<<grid, 256>>>(m, idx, x, out);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; it++) gather<POLICY><<<grid, 256>>>(m, idx, x, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("policy %d (%s): %.4f ms  %.1f G gathers/s\n", POLICY, name, ms, m / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 2000000, m = argc > 2 ? atoll(argv[2]) : 52000000;
    std::vector<int> h((size_t)m);
    std::mt19937_64 g(7);
    for (long long k = 0; k < m; k++) h[(size_t)k] = (int)(g() % (unsigned long long)n);
    int *idx; double *x, *out;
    CK(hipMalloc(&idx, sizeof(int) * (size_t)m)); CK(hipMalloc(&x, sizeof(double) * (size_t)n)); CK(hipMalloc(&out, sizeof(double) * 256 * 256 * 32));
    CK(hipMemcpy(idx, h.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, sizeof(double) * (size_t)n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("x: %lld doubles (%.1f MB), %lld random gathers\n", n, n * 8 / 1e6, m);
    run<0>("plain", m, idx, x, out, e0, e1);
    run<1>("__builtin_nontemporal_load", m, idx, x, out, e0, e1);
    run<2>("sc0", m, idx, x, out, e0, e1);
    run<3>("sc1", m, idx, x, out, e0, e1);
    run<4>("sc0 sc1", m, idx, x, out, e0, e1);
    run<5>("nt", m, idx, x, out, e0, e1);
    run<6>("sc0 nt", m, idx, x, out, e0, e1);
    run<7>("sc1 nt", m, idx, x, out, e0, e1);
    run<8>("sc0 sc1 nt", m, idx, x, out, e0, e1);
    return 0;
}
