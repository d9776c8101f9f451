// runtime.hip -- device/memory/stream/timer utilities of the kernel-level C ABI (liship.h) and the
// device-side generator of the synthetic 3-D Poisson inputs.
#include <time.h>
#include "common.hpp"
#include "liship.h"
#include <string.h>

extern "C" int liship_device_count(int *count) { HIP_TRY(hipGetDeviceCount(count)); return 0; }
extern "C" int liship_set_device(int device) { HIP_TRY(hipSetDevice(device)); return 0; }
extern "C" int liship_get_device(int *device) { HIP_TRY(hipGetDevice(device)); return 0; }

extern "C" int liship_device_name(char *buf, int buflen)
{
    if (!buf || buflen < 1) return LISHIP_ERR_ARG;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
    buf[buflen - 1] = 0;
    return 0;
}

extern "C" int liship_malloc(void **dptr, size_t bytes)
{
    const hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) { (void)hipGetLastError(); *dptr = nullptr; }   // the caller may retry: do not leave the code for the next launch check
    return (int)e;
}
extern "C" int liship_free(void *dptr) { if (dptr) HIP_TRY(hipFree(dptr)); return 0; }
extern "C" int liship_memset(void *dptr, int byte, size_t bytes, void *stream)
{ if (bytes) HIP_TRY(hipMemsetAsync(dptr, byte, bytes, as_stream(stream))); return 0; }
extern "C" int liship_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{ if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream))); return 0; }
extern "C" int liship_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{ if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream))); return 0; }
extern "C" int liship_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{ if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream))); return 0; }
extern "C" int liship_stream_create(void **stream)
{ hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = s; return 0; }
extern "C" int liship_stream_destroy(void *stream) { if (stream) HIP_TRY(hipStreamDestroy(as_stream(stream))); return 0; }
// Watchdog for jobs whose streams carry collectives (host/lis_comm.c sets it when an RCCL communicator of several ranks is formed): a stream that waits for a peer
// that never arrives would block hipStreamSynchronize for ever.  With a limit set, the wait polls hipStreamQuery (spinning for the first 2 ms: the folds of a
// Krylov loop come back in tens of microseconds; then 100 us naps) and returns LISHIP_ERR_TIMEOUT when the stream has not drained within the limit.
static double g_sync_limit_s = 0.0;
extern "C" int liship_set_sync_timeout(double seconds) { g_sync_limit_s = seconds > 0.0 ? seconds : 0.0; return 0; }
extern "C" int liship_stream_synchronize(void *stream)
{
    if (g_sync_limit_s <= 0.0) { HIP_TRY(hipStreamSynchronize(as_stream(stream))); return 0; }
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipStreamQuery(as_stream(stream));
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return (int)e;
        clock_gettime(CLOCK_MONOTONIC, &t);
        const double el = (double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec);
        if (el > g_sync_limit_s) return LISHIP_ERR_TIMEOUT;
        if (el > 2e-3) { struct timespec nap = {0, 100000}; nanosleep(&nap, nullptr); }
    }
}
extern "C" int liship_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return 0; }

// hipGraph capture of a batch of Krylov iterations (host/lis_solver.c: dev_loop_run): small systems are bound by the
// launch rate of their 5-14 kernels per iteration, and a replayed graph submits them without the per-launch host work.
// Thread-local capture: a call that may not be captured (an allocation, a synchronisation) fails the capture instead
// of being recorded wrongly, and the caller falls back to plain launches.
extern "C" int liship_graph_capture_begin(void *stream)
{ HIP_TRY(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal)); return 0; }
extern "C" int liship_graph_capture_end(void *stream, void **exec)
{
    hipGraph_t g = nullptr;
    hipError_t rc = hipStreamEndCapture(as_stream(stream), &g);          // always ends the capture, also a failed one
    if (rc != hipSuccess || !g) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); return (int)(rc != hipSuccess ? rc : hipErrorUnknown); }
    if (!exec) { (void)hipGraphDestroy(g); return 0; }                    // the caller gave up on this capture
    hipGraphExec_t e = nullptr;
    rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (rc != hipSuccess) { (void)hipGetLastError(); return (int)rc; }
    *exec = e;
    return 0;
}
extern "C" int liship_graph_launch(void *exec, void *stream)
{ if (!exec) return LISHIP_ERR_ARG; HIP_TRY(hipGraphLaunch((hipGraphExec_t)exec, as_stream(stream))); return 0; }
extern "C" int liship_graph_destroy(void *exec) { if (exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)exec)); return 0; }

// page-locked host memory: the 8-32 B results of the reductions come back through it (a D2H into pageable memory
// is staged by the runtime and costs 10-20 us more per host synchronisation)
extern "C" int liship_malloc_host(void **p, size_t bytes) { HIP_TRY(hipHostMalloc(p, bytes, hipHostMallocDefault)); return 0; }
extern "C" int liship_free_host(void *p) { if (p) HIP_TRY(hipHostFree(p)); return 0; }

// stream-ordering events (halo exchange on a second stream overlapped with the interior rows of the product)
extern "C" int liship_event_create(void **ev)
{
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return 0;
}
extern "C" int liship_event_destroy(void *ev) { if (ev) HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(ev))); return 0; }
extern "C" int liship_event_synchronize(void *ev) { HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(ev))); return 0; }
extern "C" int liship_event_record(void *ev, void *stream) { HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), as_stream(stream))); return 0; }
extern "C" int liship_stream_wait_event(void *stream, void *ev) { HIP_TRY(hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(ev), 0)); return 0; }

struct liship_timer { hipEvent_t a, b; };
extern "C" int liship_timer_create(void **timer)
{
    liship_timer *t = new liship_timer();
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { delete t; return (int)e; }
    *timer = t;
    return 0;
}
extern "C" int liship_timer_destroy(void *timer)
{
    liship_timer *t = static_cast<liship_timer *>(timer);
    if (!t) return 0;
    (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b);
    delete t;
    return 0;
}
extern "C" int liship_timer_start(void *timer, void *stream)
{ HIP_TRY(hipEventRecord(static_cast<liship_timer *>(timer)->a, as_stream(stream))); return 0; }
extern "C" int liship_timer_stop(void *timer, void *stream)
{ HIP_TRY(hipEventRecord(static_cast<liship_timer *>(timer)->b, as_stream(stream))); return 0; }
extern "C" int liship_timer_elapsed_ms(void *timer, float *ms)
{
    liship_timer *t = static_cast<liship_timer *>(timer);
    HIP_TRY(hipEventSynchronize(t->b));
    HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
    return 0;
}
// ---- the box's streaming yardstick: workgroup b reads READS consecutive tiles of 512 doubles (src[(b * READS + k) * 512 + ...]: ONE read stream that advances READS
// times as fast as the write stream) and writes their sum to dst[b * 512 + ...]; 16 B per lane, nontemporal both ways -- the traffic shape of the products
// (13 : 1 is the CSR product's read : write ratio on SURVEY 8d's bytes) without a single gather or index.  What this kernel reaches on a box is what "the HBM
// roofline" means on that box; bench.py prints it beside the product's fraction.  (READS separate streams n doubles apart -- the first form of this kernel --
// reach LESS than the CSR product itself: 5.8 against 6.4 TB/s on the same box.)
namespace {
template <int READS>
__global__ __launch_bounds__(256) void stream_sum_kernel(size_t ntiles, const double *__restrict__ src, double *__restrict__ dst)
{
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {               // persistent workgroups walking the tiles grid-stride (gridDim = 0: one tile per workgroup)
        const double *s = src + t * 512 * READS + (size_t)threadIdx.x * 2;
        v2f64 v[READS];
#pragma unroll
        for (int k = 0; k < READS; k++) v[k] = load_stream(reinterpret_cast<const v2f64 *>(s + (size_t)k * 512));
        v2f64 a = v[0];
#pragma unroll
        for (int k = 1; k < READS; k++) { a.x += v[k].x; a.y += v[k].y; }
        store_stream(reinterpret_cast<v2f64 *>(dst + t * 512 + (size_t)threadIdx.x * 2), a);
    }
}
}
// wgs_per_cu: 0 = one workgroup per tile, else that many persistent workgroups per CU (256 CUs)
extern "C" int liship_stream_yardstick(int reads, size_t n, const double *src, double *dst, int wgs_per_cu, void *stream)
{
    if (!src || !dst || (n & 511) || n == 0 || wgs_per_cu < 0) return LISHIP_ERR_ARG;
    const size_t ntiles = n / 512;
    const unsigned grid = wgs_per_cu > 0 ? (unsigned)(256 * wgs_per_cu) : (unsigned)ntiles;
    hipStream_t st = as_stream(stream);
    switch (reads) {
    case 1:  stream_sum_kernel<1><<<grid, 256, 0, st>>>(ntiles, src, dst); break;
    case 2:  stream_sum_kernel<2><<<grid, 256, 0, st>>>(ntiles, src, dst); break;
    case 4:  stream_sum_kernel<4><<<grid, 256, 0, st>>>(ntiles, src, dst); break;
    case 8:  stream_sum_kernel<8><<<grid, 256, 0, st>>>(ntiles, src, dst); break;
    case 13: stream_sum_kernel<13><<<grid, 256, 0, st>>>(ntiles, src, dst); break;
    default: return LISHIP_ERR_ARG;
    }
    LAUNCH_CHECK();
    return 0;
}

extern "C" const char *liship_error_string(int code)
{
    if (code == 0) return "success";
    if (code == LISHIP_ERR_ARG) return "liship: invalid argument";
    return hipGetErrorString((hipError_t)code);
}

// ------------------------------------------------------------------------------------------------
// 3-D 7-point Poisson rows [is,ie) generated in place (no host matrix, no PCIe).  Row g = i*m*n + j*n + k
// has -1 for each in-range neighbour and +6 on the diagonal; entry order as test/test3.c:114-127
// (-mn,+mn,-n,+n,-1,+1,diag) or ascending column (test/spmvtest3.c:192-195).
namespace {

struct Grid { int l, m, n; long long mn; };

// number of matrix entries in global rows [0,g): 7g minus the neighbours cut off by the six faces
__host__ __device__ inline long long entries_before(const Grid &G, long long g)
{
    const long long mn = G.mn, nn = G.n;
    const long long planes = g / mn, rem = g % mn;
    long long cut = 0;
    cut += (g < mn ? g : mn);                                           // i == 0     : no -mn
    cut += (g > (long long)(G.l - 1) * mn ? g - (long long)(G.l - 1) * mn : 0);   // i == l-1   : no +mn
    cut += planes * nn + (rem < nn ? rem : nn);                         // j == 0     : no -n
    cut += planes * nn + (rem > (long long)(G.m - 1) * nn ? rem - (long long)(G.m - 1) * nn : 0); // j == m-1
    cut += (g + nn - 1) / nn;                                           // k == 0     : no -1
    cut += g / nn;                                                      // k == n-1   : no +1
    return 7 * g - cut;
}

__global__ __launch_bounds__(256)
void poisson3d_kernel(Grid G, int is, int ie, int sorted, int nlow,
                      int *__restrict__ ptr, int *__restrict__ idx, double *__restrict__ val)
{
    const int nloc = ie - is;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > nloc) return;
    const long long g = (long long)is + r;
    const long long base = entries_before(G, is);
    const int start = (int)(entries_before(G, g) - base);
    ptr[r] = start;
    if (r == nloc) return;
    const int mn = (int)G.mn;
    const int i = (int)(g / mn), rem = (int)(g % mn), j = rem / G.n, k = rem % G.n;
    // neighbour offsets in the requested order; 0 marks the diagonal
    int offs[7];
    if (sorted) { offs[0] = -mn; offs[1] = -G.n; offs[2] = -1; offs[3] = 0; offs[4] = 1; offs[5] = G.n; offs[6] = mn; }
    else        { offs[0] = -mn; offs[1] = mn; offs[2] = -G.n; offs[3] = G.n; offs[4] = -1; offs[5] = 1; offs[6] = 0; }
    int w = start;
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int o = offs[t];
        bool present;
        if (o == -mn) present = i > 0;
        else if (o == mn) present = i < G.l - 1;
        else if (o == -G.n) present = j > 0;
        else if (o == G.n) present = j < G.m - 1;
        else if (o == -1) present = k > 0;
        else if (o == 1) present = k < G.n - 1;
        else present = true;
        if (!present) continue;
        const long long c = g + o;
        int lc;
        if (c >= is && c < ie) lc = (int)(c - is);
        else if (c < is)       lc = nloc + (int)(c - ((long long)is - mn));     // lower ghost plane
        else                   lc = nloc + nlow + (int)(c - ie);                 // upper ghost plane
        idx[w] = lc;
        val[w] = (o == 0) ? 6.0 : -1.0;
        w++;
    }
}

__global__ __launch_bounds__(256)
void poisson3d_rhs_kernel(Grid G, int is, int ie, double *__restrict__ b)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= ie - is) return;
    const long long g = (long long)is + r;
    const int mn = (int)G.mn;
    const int i = (int)(g / mn), rem = (int)(g % mn), j = rem / G.n, k = rem % G.n;
    const int nb = (i > 0) + (i < G.l - 1) + (j > 0) + (j < G.m - 1) + (k > 0) + (k < G.n - 1);
    b[r] = 6.0 - (double)nb;
}

bool slab_ok(int l, int m, int n, int is, int ie)
{
    if (l < 1 || m < 1 || n < 1) return false;
    const long long mn = (long long)m * n, gn = mn * l;
    if (gn > 0x7fffffffLL || is < 0 || ie < is || ie > gn) return false;
    // ghost numbering above assumes whole planes per slab unless the slab is the whole grid
    if ((is % mn) != 0 || (ie % mn) != 0) return false;
    return true;
}

} // namespace

extern "C" long long liship_poisson3d_nnz(int l, int m, int n, int is, int ie)
{
    if (!slab_ok(l, m, n, is, ie)) return -1;
    Grid G{l, m, n, (long long)m * n};
    return entries_before(G, ie) - entries_before(G, is);
}

extern "C" int liship_poisson3d_csr(int l, int m, int n, int is, int ie, int sorted,
                                    int *ptr, int *index, double *value, void *stream)
{
    if (!slab_ok(l, m, n, is, ie)) return LISHIP_ERR_ARG;
    Grid G{l, m, n, (long long)m * n};
    if (entries_before(G, ie) - entries_before(G, is) > 0x7fffffffLL) return LISHIP_ERR_ARG;
    const int nloc = ie - is, nlow = is > 0 ? (int)G.mn : 0;
    poisson3d_kernel<<<(nloc + 1 + 255) / 256, 256, 0, as_stream(stream)>>>(G, is, ie, sorted, nlow, ptr, index, value);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_poisson3d_rhs(int l, int m, int n, int is, int ie, double *b, void *stream)
{
    if (!slab_ok(l, m, n, is, ie)) return LISHIP_ERR_ARG;
    if (ie == is) return 0;
    Grid G{l, m, n, (long long)m * n};
    poisson3d_rhs_kernel<<<(ie - is + 255) / 256, 256, 0, as_stream(stream)>>>(G, is, ie, b);
    LAUNCH_CHECK();
    return 0;
}
