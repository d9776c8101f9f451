// dia_pad_proto.hip -- experiment: does the distance between the diagonals of a DIA matrix (and the jagged columns of an ELL matrix) in HBM matter?
// The reference's layouts put diagonal d at value[d * n + i]: at 512^3 rows that is a stride of exactly 2^30 B, at 256^3 2^27 B -- seven read streams that
// advance in lockstep through the SAME channel / bank bits of the HBM address map.  This times the 7-point DIA product with a leading dimension of n + pad rows.
//   hipcc --offload-arch=gfx950 -O3 -o dia_pad_proto tools/proto/dia_pad_proto.hip && ./dia_pad_proto [G=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2f64 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int strip_unit(int w, int n, int plane)
{
    if (plane <= 0) return w;
    const int full = (n / plane) * plane;
    if (w >= full) return w;
    const int sb = plane >> 3, xcd = w & 7, slot = w >> 3, pl = slot / sb;
    return pl * plane + xcd * sb + (slot - pl * sb);
}

template <int NND>
__global__ __launch_bounds__(256) void dia_kernel(int n, size_t ld, const int *__restrict__ off, const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, int plane)
{
    const int bid = strip_unit((int)blockIdx.x, (int)gridDim.x, plane);
    const int r = (bid * 256 + (int)threadIdx.x) * 2;
    if (r >= n) return;
    double a0 = 0.0, a1 = 0.0;
    v2f64 v[NND];
    double x0[NND], x1[NND];
    bool k0[NND], k1[NND];
#pragma unroll
    for (int d = 0; d < NND; d++) {
        v[d] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(val + (size_t)d * ld + (size_t)r));
        const int o = off[d], c0 = r + o, c1 = r + 1 + o;
        k0[d] = c0 >= 0 && c0 < n; k1[d] = c1 >= 0 && c1 < n;
        x0[d] = x[k0[d] ? c0 : r]; x1[d] = x[k1[d] ? c1 : r];
    }
#pragma unroll
    for (int d = 0; d < NND; d++) { const double t0 = v[d].x * x0[d], t1 = v[d].y * x1[d]; a0 += k0[d] ? t0 : 0.0; a1 += k1[d] ? t1 : 0.0; }
    v2f64 o; o.x = a0; o.y = a1;
    __builtin_nontemporal_store(o, reinterpret_cast<v2f64 *>(y + r));
}

// ELL: value[j * ld + i], index[j * ld + i]
template <int W>
__global__ __launch_bounds__(256) void ell_kernel(int n, size_t ld, const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, int plane)
{
    typedef int v2i32 __attribute__((ext_vector_type(2)));
    const int bid = strip_unit((int)blockIdx.x, (int)gridDim.x, plane);
    const int r = (bid * 256 + (int)threadIdx.x) * 2;
    if (r >= n) return;
    double a0 = 0.0, a1 = 0.0;
    v2f64 v[W]; v2i32 c[W]; double x0[W], x1[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(val + (size_t)j * ld + (size_t)r));
        c[j] = __builtin_nontemporal_load(reinterpret_cast<const v2i32 *>(idx + (size_t)j * ld + (size_t)r));
    }
#pragma unroll
    for (int j = 0; j < W; j++) { x0[j] = x[c[j].x]; x1[j] = x[c[j].y]; }
#pragma unroll
    for (int j = 0; j < W; j++) { a0 += v[j].x * x0[j]; a1 += v[j].y * x1[j]; }
    v2f64 o; o.x = a0; o.y = a1;
    __builtin_nontemporal_store(o, reinterpret_cast<v2f64 *>(y + r));
}

// TILED: the NND diagonals of a workgroup's 512 rows are consecutive in HBM (val_t[(b * NND + d) * 512 + j]): one read stream instead of NND
template <int NND>
__global__ __launch_bounds__(256) void dia_tiled_kernel(int n, const int *__restrict__ off, const double *__restrict__ val, const double *__restrict__ x,
                                                        double *__restrict__ y, int plane)
{
    const int bid = strip_unit((int)blockIdx.x, (int)gridDim.x, plane);
    const int r = (bid * 256 + (int)threadIdx.x) * 2;
    if (r >= n) return;
    const double *vt = val + (size_t)bid * 512 * NND + (size_t)threadIdx.x * 2;
    double a0 = 0.0, a1 = 0.0;
    v2f64 v[NND];
    double x0[NND], x1[NND];
    bool k0[NND], k1[NND];
#pragma unroll
    for (int d = 0; d < NND; d++) {
        v[d] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(vt + (size_t)d * 512));
        const int o = off[d], c0 = r + o, c1 = r + 1 + o;
        k0[d] = c0 >= 0 && c0 < n; k1[d] = c1 >= 0 && c1 < n;
        x0[d] = x[k0[d] ? c0 : r]; x1[d] = x[k1[d] ? c1 : r];
    }
#pragma unroll
    for (int d = 0; d < NND; d++) { const double t0 = v[d].x * x0[d], t1 = v[d].y * x1[d]; a0 += k0[d] ? t0 : 0.0; a1 += k1[d] ? t1 : 0.0; }
    v2f64 o; o.x = a0; o.y = a1;
    __builtin_nontemporal_store(o, reinterpret_cast<v2f64 *>(y + r));
}
template <int W>
__global__ __launch_bounds__(256) void ell_tiled_kernel(int n, const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x,
                                                        double *__restrict__ y, int plane)
{
    typedef int v2i32 __attribute__((ext_vector_type(2)));
    const int bid = strip_unit((int)blockIdx.x, (int)gridDim.x, plane);
    const int r = (bid * 256 + (int)threadIdx.x) * 2;
    if (r >= n) return;
    const double *vt = val + (size_t)bid * 512 * W + (size_t)threadIdx.x * 2;
    const int *it = idx + (size_t)bid * 512 * W + (size_t)threadIdx.x * 2;
    double a0 = 0.0, a1 = 0.0;
    v2f64 v[W]; v2i32 c[W]; double x0[W], x1[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(vt + (size_t)j * 512));
        c[j] = __builtin_nontemporal_load(reinterpret_cast<const v2i32 *>(it + (size_t)j * 512));
    }
#pragma unroll
    for (int j = 0; j < W; j++) { x0[j] = x[c[j].x]; x1[j] = x[c[j].y]; }
#pragma unroll
    for (int j = 0; j < W; j++) { a0 += v[j].x * x0[j]; a1 += v[j].y * x1[j]; }
    v2f64 o; o.x = a0; o.y = a1;
    __builtin_nontemporal_store(o, reinterpret_cast<v2f64 *>(y + r));
}
__global__ void tile_f64(int n, size_t ld, int w, const double *src, double *dst)      // dst[(b * w + d) * 512 + j] = src[d * ld + b * 512 + j]
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n) return;
    const size_t b = i / 512, j = i % 512;
    for (int d = 0; d < w; d++) dst[(b * w + d) * 512 + j] = src[(size_t)d * ld + i];
}
__global__ void tile_i32(int n, size_t ld, int w, const int *src, int *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n) return;
    const size_t b = i / 512, j = i % 512;
    for (int d = 0; d < w; d++) dst[(b * w + d) * 512 + j] = src[(size_t)d * ld + i];
}

__global__ void fill_dia(int n, size_t ld, int G, double *val)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n) return;
    for (int d = 0; d < 7; d++) val[(size_t)d * ld + i] = d == 3 ? 6.0 + 1e-3 * (double)(i % 97) : -1.0 - 1e-3 * (double)((i + d) % 89);
}
__global__ void fill_ell(int n, size_t ld, int G, int *idx, double *val)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n) return;
    const long long offs[7] = {-(long long)G * G, -G, -1, 0, 1, G, (long long)G * G};
    for (int j = 0; j < 7; j++) {
        long long c = (long long)i + offs[j];
        const bool ok = c >= 0 && c < n;
        idx[(size_t)j * ld + i] = ok ? (int)c : (int)i;
        val[(size_t)j * ld + i] = ok ? (j == 3 ? 6.0 : -1.0 - 1e-3 * (double)((i + j) % 89)) : 0.0;
    }
}
__global__ void fill_x(int n, double *x)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n) { const double t = (double)i * 0.6180339887498949; x[i] = t - floor(t) - 0.5; }
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 512;
    const int n = G * G * G;
    const int plane_wg = (G * G) / 512;          // workgroups per grid plane (512 rows per workgroup)
    const size_t pads[] = {0, 16, 1024};
    double *x, *y; int *off;
    CK(hipMalloc(&x, sizeof(double) * ((size_t)n + 16))); CK(hipMalloc(&y, sizeof(double) * ((size_t)n + 16))); CK(hipMalloc(&off, 7 * sizeof(int)));
    const int hoff[7] = {-G * G, -G, -1, 0, 1, G, G * G};
    CK(hipMemcpy(off, hoff, sizeof(hoff), hipMemcpyHostToDevice));
    fill_x<<<(n + 255) / 256, 256>>>(n, x);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> yref((size_t)4096), yh((size_t)4096);
    for (int fmt = 0; fmt < 2; fmt++) {
        for (size_t pi = 0; pi < sizeof(pads) / sizeof(pads[0]); pi++) {
            const size_t ld = (size_t)n + pads[pi];            // in elements (pads are even: 16 B alignment of every diagonal kept)
            double *val; int *idx = nullptr;
            CK(hipMalloc(&val, sizeof(double) * 7 * ld));
            if (fmt == 1) { CK(hipMalloc(&idx, sizeof(int) * 7 * ld)); fill_ell<<<(n + 255) / 256, 256>>>(n, ld, G, idx, val); }
            else fill_dia<<<(n + 255) / 256, 256>>>(n, ld, G, val);
            CK(hipDeviceSynchronize());
            const int grid = (n / 2 + 255) / 256;
            for (int strips = 1; strips >= 0; strips--) {
                const int pl = strips ? plane_wg : 0;
                float best = 1e30f, sum = 0;
                for (int rep = 0; rep < 3; rep++) {
                    for (int w = 0; w < 3; w++) { if (fmt) ell_kernel<7><<<grid, 256>>>(n, ld, idx, val, x, y, pl); else dia_kernel<7><<<grid, 256>>>(n, ld, off, val, x, y, pl); }
                    CK(hipEventRecord(e0));
                    for (int it = 0; it < 10; it++) { if (fmt) ell_kernel<7><<<grid, 256>>>(n, ld, idx, val, x, y, pl); else dia_kernel<7><<<grid, 256>>>(n, ld, off, val, x, y, pl); }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
                    if (ms < best) best = ms;
                    sum += ms;
                }
                const double B = fmt ? 100.0 * n : 72.0 * n;
                CK(hipMemcpy(yh.data(), y + n / 2, sizeof(double) * 4096, hipMemcpyDeviceToHost));
                if (pi == 0 && strips == 1) yref = yh;
                bool same = true; for (int i = 0; i < 4096; i++) same = same && yh[i] == yref[i];
                printf("%s G=%d pad=%8zu elements strips=%d: best %.4f ms mean %.4f ms  frac(best) %.4f  frac(mean) %.4f %s\n", fmt ? "ELL" : "DIA", G, pads[pi], strips, best, sum / 3,
                       B / (best * 1e-3) / 8e12, B / (sum / 3 * 1e-3) / 8e12, same ? "" : "Y DIFFERS");
            }
            if (pi == 0) {                                   // the tiled layout of the same matrix
                double *vt; int *it = nullptr;
                CK(hipMalloc(&vt, sizeof(double) * 7 * ld));
                tile_f64<<<(n + 255) / 256, 256>>>(n, ld, 7, val, vt);
                if (fmt == 1) { CK(hipMalloc(&it, sizeof(int) * 7 * ld)); tile_i32<<<(n + 255) / 256, 256>>>(n, ld, 7, idx, it); }
                CK(hipDeviceSynchronize());
                for (int strips = 1; strips >= 0; strips--) {
                    const int pl = strips ? plane_wg : 0;
                    float best = 1e30f, sum = 0;
                    for (int rep = 0; rep < 3; rep++) {
                        for (int w = 0; w < 3; w++) { if (fmt) ell_tiled_kernel<7><<<grid, 256>>>(n, it, vt, x, y, pl); else dia_tiled_kernel<7><<<grid, 256>>>(n, off, vt, x, y, pl); }
                        CK(hipEventRecord(e0));
                        for (int k = 0; k < 10; k++) { if (fmt) ell_tiled_kernel<7><<<grid, 256>>>(n, it, vt, x, y, pl); else dia_tiled_kernel<7><<<grid, 256>>>(n, off, vt, x, y, pl); }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
                        if (ms < best) best = ms;
                        sum += ms;
                    }
                    const double B = fmt ? 100.0 * n : 72.0 * n;
                    CK(hipMemcpy(yh.data(), y + n / 2, sizeof(double) * 4096, hipMemcpyDeviceToHost));
                    bool same = true; for (int i = 0; i < 4096; i++) same = same && yh[i] == yref[i];
                    printf("%s G=%d TILED (512-row tiles) strips=%d: best %.4f ms mean %.4f ms  frac(best) %.4f  frac(mean) %.4f %s\n", fmt ? "ELL" : "DIA", G, strips, best, sum / 3,
                           B / (best * 1e-3) / 8e12, B / (sum / 3 * 1e-3) / 8e12, same ? "" : "Y DIFFERS");
                }
                CK(hipFree(vt)); if (it) CK(hipFree(it));
            }
            CK(hipFree(val)); if (idx) CK(hipFree(idx));
        }
    }
    return 0;
}
