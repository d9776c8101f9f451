// Prototype only (not part of the library): the 7-point constant-coefficient product as a z-MARCHING kernel -- each workgroup owns a TX x TY tile of a plane and walks
// ZSEG planes; a plane's tile (+ halo) is loaded ONCE (coalesced 16 B loads, D planes ahead, held in registers), written to LDS, and the +-1 / +-NX neighbours come from
// LDS, the +-NX*NY ones from the registers of the planes before and after.  Interior rows only (faces compute garbage): what could the headline kernel gain?
//   hipcc --offload-arch=gfx950 -O3 -o march_proto march_proto.hip && ./march_proto [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef double v2f64u __attribute__((ext_vector_type(2), aligned(8)));
#define CK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } } while (0)

struct Coef { double v[7]; };

// TX = 128 columns (64 lanes x pairs), TY lines = 4 waves x LPW lines per wave
template <int LPW, int D, bool XCD>
__global__ __launch_bounds__(256)
void march_kernel(const unsigned char *__restrict__ rowpat, const double *__restrict__ x, double *__restrict__ y, int N, int zseg, Coef C, int tiles_x, int tiles_y)
{
    constexpr int TX = 128, TY = 4 * LPW, LX = TX + 4;               // LDS line: [1 pad][left halo][TX][right halo][1 pad] -> own pair at 2 + 2 cp: 16 B aligned
    __shared__ __attribute__((aligned(16))) double buf[2][(TY + 2) * LX];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    int wg = blockIdx.x;
    const int ntile = tiles_x * tiles_y, nseg = (N + zseg - 1) / zseg;
    if (XCD) { const int k = wg % 8, j = wg / 8, per = (ntile * nseg) / 8; wg = k * per + j; }      // each XCD a contiguous eighth of the (tile, segment) list
    const int seg = wg / ntile, t = wg - seg * ntile;                // consecutive workgroups: neighbouring tiles of one z segment
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int z0 = seg * zseg, z1 = min(N, z0 + zseg);
    const long long S = N, SO = (long long)N * N;
    const int col0 = tx * TX, line0 = ty * TY;
    // this lane's points: columns col0 + 2 lane, +1 on lines line0 + w * LPW + i
    const long long base = (long long)(line0 + w * LPW) * S + col0 + 2 * lane;
    const int total = N * N * N;
    auto clampi = [&](long long a) { return (int)(a < 0 ? 0 : (a > total - 2 ? total - 2 : a)); };
    struct Packet { v2f64 own[LPW]; v2f64 hy; double hx[LPW]; };
    auto load_packet = [&](Packet &P, int z) {
        const long long pb = (long long)z * SO + base;
#pragma unroll
        for (int i = 0; i < LPW; i++) P.own[i] = *reinterpret_cast<const v2f64u *>(x + clampi(pb + i * S));
        // y halo: wave 0 loads the line above the tile, wave 3 the line below
        if (w == 0) P.hy = *reinterpret_cast<const v2f64u *>(x + clampi((long long)z * SO + (long long)(line0 - 1) * S + col0 + 2 * lane));
        if (w == 3) P.hy = *reinterpret_cast<const v2f64u *>(x + clampi((long long)z * SO + (long long)(line0 + TY) * S + col0 + 2 * lane));
        // x halo: lane 0 the column left of the tile, lane 63 the column right of it, for the wave's lines
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.hx[i] = x[clampi((long long)z * SO + (long long)(line0 + w * LPW + i) * S + (lane == 0 ? col0 - 1 : col0 + TX))];
        }
    };
    auto store_packet = [&](const Packet &P, double *B) {
#pragma unroll
        for (int i = 0; i < LPW; i++) *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + 2 + 2 * lane) = P.own[i];
        if (w == 0) *reinterpret_cast<v2f64 *>(B + 0 * LX + 2 + 2 * lane) = P.hy;
        if (w == 3) *reinterpret_cast<v2f64 *>(B + (TY + 1) * LX + 2 + 2 * lane) = P.hy;
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int i = 0; i < LPW; i++) B[(w * LPW + i + 1) * LX + (lane == 0 ? 1 : 2 + TX)] = P.hx[i];
        }
    };
    Packet Q[D];
    v2f64 prev[LPW], cur[LPW];
    // prologue: planes z0 - 1 (registers only), z0 (LDS + registers), packets z0 + 1 .. z0 + D in flight
    {
        Packet P;
        load_packet(P, z0 - 1);
#pragma unroll
        for (int i = 0; i < LPW; i++) prev[i] = P.own[i];
        load_packet(P, z0);
        store_packet(P, buf[z0 & 1]);
#pragma unroll
        for (int i = 0; i < LPW; i++) cur[i] = P.own[i];
    }
#pragma unroll
    for (int d = 0; d < D; d++) load_packet(Q[d], z0 + 1 + d);
    __syncthreads();
    for (int zb = z0; zb < z1; zb += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int z = zb + d;
            if (z < z1) {                                                   // (uniform)
                // packet z + 1 has landed (the oldest in flight): its tile goes to the other LDS buffer
                Packet &P = Q[d];
                store_packet(P, buf[(z + 1) & 1]);
                const unsigned short two = *reinterpret_cast<const unsigned short *>(rowpat + clampi((long long)z * SO + base));
                const double *B = buf[z & 1];
#pragma unroll
                for (int i = 0; i < LPW; i++) {
                    const int li = (w * LPW + i + 1) * LX + 2 + 2 * lane;
                    const v2f64 c = *reinterpret_cast<const v2f64 *>(B + li);
                    const double l = B[li - 1], r = B[li + 2];
                    const v2f64 up = *reinterpret_cast<const v2f64 *>(B + li - LX), dn = *reinterpret_cast<const v2f64 *>(B + li + LX);
                    double s0 = 0.0, s1 = 0.0;
                    s0 += C.v[0] * prev[i].x;  s1 += C.v[0] * prev[i].y;
                    s0 += C.v[1] * up.x;       s1 += C.v[1] * up.y;
                    s0 += C.v[2] * l;          s1 += C.v[2] * c.x;
                    s0 += C.v[3] * c.x;        s1 += C.v[3] * c.y;
                    s0 += C.v[4] * c.y;        s1 += C.v[4] * r;
                    s0 += C.v[5] * dn.x;       s1 += C.v[5] * dn.y;
                    s0 += C.v[6] * P.own[i].x; s1 += C.v[6] * P.own[i].y;
                    if (two == 0xffff) { s0 = -s0; s1 = -s1; }            // (keeps the pattern bytes alive; never true)
                    v2f64 out; out.x = s0; out.y = s1;
                    __builtin_nontemporal_store(out, reinterpret_cast<v2f64 *>(y + (long long)z * SO + base + i * S));
                    prev[i] = c; 
                }
                // the next packet for this slot: plane z + 1 + D
                {
                    Packet N2;
                    load_packet(N2, z + 1 + D);
                    // rotate: cur is not needed (the centre comes from LDS); Q[d] becomes the new packet after its own values were used above
                    Q[d] = N2;
                }
                __syncthreads();
            }
        }
    }
}

// TX = 128 columns (64 lanes x pairs), TY lines = 4 waves x LPW lines per wave
template <int LPW, int D, bool XCD, int VD>
__global__ __launch_bounds__(256)
void marchv_kernel(const unsigned char *__restrict__ rowpat, const double *__restrict__ x, double *__restrict__ y, int N, int zseg, const double *__restrict__ vals, int tiles_x, int tiles_y)
{
    constexpr int TX = 128, TY = 4 * LPW, LX = TX + 4;               // LDS line: [1 pad][left halo][TX][right halo][1 pad] -> own pair at 2 + 2 cp: 16 B aligned
    __shared__ __attribute__((aligned(16))) double buf[2][(TY + 2) * LX];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    int wg = blockIdx.x;
    const int ntile = tiles_x * tiles_y, nseg = (N + zseg - 1) / zseg;
    if (XCD) { const int k = wg % 8, j = wg / 8, per = (ntile * nseg) / 8; wg = k * per + j; }      // each XCD a contiguous eighth of the (tile, segment) list
    const int seg = wg / ntile, t = wg - seg * ntile;                // consecutive workgroups: neighbouring tiles of one z segment
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int z0 = seg * zseg, z1 = min(N, z0 + zseg);
    const long long S = N, SO = (long long)N * N;
    const int col0 = tx * TX, line0 = ty * TY;
    // this lane's points: columns col0 + 2 lane, +1 on lines line0 + w * LPW + i
    const long long base = (long long)(line0 + w * LPW) * S + col0 + 2 * lane;
    const int total = N * N * N;
    auto clampi = [&](long long a) { return (int)(a < 0 ? 0 : (a > total - 2 ? total - 2 : a)); };
    struct Packet { v2f64 own[LPW]; v2f64 hy; double hx[LPW]; };
    auto load_packet = [&](Packet &P, int z) {
        const long long pb = (long long)z * SO + base;
#pragma unroll
        for (int i = 0; i < LPW; i++) P.own[i] = *reinterpret_cast<const v2f64u *>(x + clampi(pb + i * S));
        // y halo: wave 0 loads the line above the tile, wave 3 the line below
        if (w == 0) P.hy = *reinterpret_cast<const v2f64u *>(x + clampi((long long)z * SO + (long long)(line0 - 1) * S + col0 + 2 * lane));
        if (w == 3) P.hy = *reinterpret_cast<const v2f64u *>(x + clampi((long long)z * SO + (long long)(line0 + TY) * S + col0 + 2 * lane));
        // x halo: lane 0 the column left of the tile, lane 63 the column right of it, for the wave's lines
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.hx[i] = x[clampi((long long)z * SO + (long long)(line0 + w * LPW + i) * S + (lane == 0 ? col0 - 1 : col0 + TX))];
        }
    };
    auto store_packet = [&](const Packet &P, double *B) {
#pragma unroll
        for (int i = 0; i < LPW; i++) *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + 2 + 2 * lane) = P.own[i];
        if (w == 0) *reinterpret_cast<v2f64 *>(B + 0 * LX + 2 + 2 * lane) = P.hy;
        if (w == 3) *reinterpret_cast<v2f64 *>(B + (TY + 1) * LX + 2 + 2 * lane) = P.hy;
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int i = 0; i < LPW; i++) B[(w * LPW + i + 1) * LX + (lane == 0 ? 1 : 2 + TX)] = P.hx[i];
        }
    };
    Packet Q[D];
    v2f64 prev[LPW], cur[LPW];
    v2f64 VQ[VD][7][LPW];
    auto load_vals = [&](v2f64 (&V)[7][LPW], int z) {
        const long long pb = (long long)z * SO + base;
#pragma unroll
        for (int u = 0; u < 7; u++)
#pragma unroll
            for (int i = 0; i < LPW; i++) V[u][i] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(vals + (size_t)u * total + clampi(pb + i * S)));
    };
    // prologue: planes z0 - 1 (registers only), z0 (LDS + registers), packets z0 + 1 .. z0 + D in flight
    {
        Packet P;
        load_packet(P, z0 - 1);
#pragma unroll
        for (int i = 0; i < LPW; i++) prev[i] = P.own[i];
        load_packet(P, z0);
        store_packet(P, buf[z0 & 1]);
#pragma unroll
        for (int i = 0; i < LPW; i++) cur[i] = P.own[i];
    }
#pragma unroll
    for (int d = 0; d < D; d++) load_packet(Q[d], z0 + 1 + d);
#pragma unroll
    for (int d = 0; d < VD; d++) load_vals(VQ[d], z0 + d);
    __syncthreads();
    static_assert(VD == D, "one rotation");
    for (int zb = z0; zb < z1; zb += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int z = zb + d;
            v2f64 (&V)[7][LPW] = VQ[d];
            if (z < z1) {                                                   // (uniform)
                // packet z + 1 has landed (the oldest in flight): its tile goes to the other LDS buffer
                Packet &P = Q[d];
                store_packet(P, buf[(z + 1) & 1]);
                const unsigned short two = *reinterpret_cast<const unsigned short *>(rowpat + clampi((long long)z * SO + base));
                const double *B = buf[z & 1];
#pragma unroll
                for (int i = 0; i < LPW; i++) {
                    const int li = (w * LPW + i + 1) * LX + 2 + 2 * lane;
                    const v2f64 c = *reinterpret_cast<const v2f64 *>(B + li);
                    const double l = B[li - 1], r = B[li + 2];
                    const v2f64 up = *reinterpret_cast<const v2f64 *>(B + li - LX), dn = *reinterpret_cast<const v2f64 *>(B + li + LX);
                    double s0 = 0.0, s1 = 0.0;
                    s0 += V[0][i].x * prev[i].x;  s1 += V[0][i].y * prev[i].y;
                    s0 += V[1][i].x * up.x;       s1 += V[1][i].y * up.y;
                    s0 += V[2][i].x * l;          s1 += V[2][i].y * c.x;
                    s0 += V[3][i].x * c.x;        s1 += V[3][i].y * c.y;
                    s0 += V[4][i].x * c.y;        s1 += V[4][i].y * r;
                    s0 += V[5][i].x * dn.x;       s1 += V[5][i].y * dn.y;
                    s0 += V[6][i].x * P.own[i].x; s1 += V[6][i].y * P.own[i].y;
                    if (two == 0xffff) { s0 = -s0; s1 = -s1; }            // (keeps the pattern bytes alive; never true)
                    v2f64 out; out.x = s0; out.y = s1;
                    __builtin_nontemporal_store(out, reinterpret_cast<v2f64 *>(y + (long long)z * SO + base + i * S));
                    prev[i] = c; 
                }
                // the next packet for this slot: plane z + 1 + D
                {
                    Packet N2;
                    load_packet(N2, z + 1 + D);
                    // rotate: cur is not needed (the centre comes from LDS); Q[d] becomes the new packet after its own values were used above
                    Q[d] = N2;
                }
                load_vals(VQ[d], z + VD);
                __syncthreads();
            }
        }
    }
}

// reference form: what the library runs today, stripped to the interior: lane pair, seven 16 B gathers
__global__ __launch_bounds__(256)
void gather_kernel(const unsigned char *__restrict__ rowpat, const double *__restrict__ x, double *__restrict__ y, int N, Coef C)
{
    const long long SO = (long long)N * N, total = SO * N;
    const long long r = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (r + 1 >= total) return;
    auto cl = [&](long long a) { return a < 0 ? 0 : (a > total - 2 ? total - 2 : a); };
    const long long off[7] = {-SO, -N, -1, 0, 1, N, SO};
    v2f64 xx[7];
#pragma unroll
    for (int u = 0; u < 7; u++) xx[u] = *reinterpret_cast<const v2f64u *>(x + cl(r + off[u]));
    const unsigned short two = *reinterpret_cast<const unsigned short *>(rowpat + r);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int u = 0; u < 7; u++) { s0 += C.v[u] * xx[u].x; s1 += C.v[u] * xx[u].y; }
    if (two == 0xffff) { s0 = -s0; s1 = -s1; }
    v2f64 out; out.x = s0; out.y = s1;
    __builtin_nontemporal_store(out, reinterpret_cast<v2f64 *>(y + r));
}

template <typename F> static float timeit(F f, int warm, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < warm; i++) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 512;
    const size_t n = (size_t)N * N * N;
    std::vector<double> hx(n);
    for (size_t i = 0; i < n; i++) hx[i] = fmod((double)i * 0.6180339887498949, 1.0) - 0.5;
    double *x, *y, *y2; unsigned char *pat;
    CK(hipMalloc(&x, 8 * n + 64)); CK(hipMalloc(&y, 8 * n + 64)); CK(hipMalloc(&y2, 8 * n + 64)); CK(hipMalloc(&pat, n + 64));
    CK(hipMemcpy(x, hx.data(), 8 * n, hipMemcpyHostToDevice)); CK(hipMemset(pat, 13, n + 64)); CK(hipMemset(y, 0, 8 * n)); CK(hipMemset(y2, 0, 8 * n));
    Coef C = {{-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0}};
    const double bytes = 17.0 * n;
    float ms = timeit([&] { gather_kernel<<<(unsigned)((n / 2 + 255) / 256), 256>>>(pat, x, y2, N, C); }, 20, 50);
    printf("gather (seven 16 B gathers per lane pair, linear rows): %.4f ms  %.3f of 8 TB/s on 17 B/row\n", ms, bytes / ms / 1e6 / 8000);
    std::vector<double> h1(n), h2(n);
    CK(hipMemcpy(h2.data(), y2, 8 * n, hipMemcpyDeviceToHost));
#define RUN(LPW, D, XCD, ZSEG) do { \
        const int tiles_x = N / 128, tiles_y = N / (4 * LPW), nseg = (N + ZSEG - 1) / ZSEG; \
        if (N % 128 || N % (4 * LPW)) break; \
        CK(hipMemset(y, 0, 8 * n)); \
        ms = timeit([&] { march_kernel<LPW, D, XCD><<<tiles_x * tiles_y * nseg, 256>>>(pat, x, y, N, ZSEG, C, tiles_x, tiles_y); }, 10, 30); \
        CK(hipGetLastError()); \
        CK(hipMemcpy(h1.data(), y, 8 * n, hipMemcpyDeviceToHost)); \
        size_t bad = 0, checked = 0; \
        for (int z = 1; z < N - 1; z += 7) for (int yy = 1; yy < N - 1; yy += 5) for (int xx = 1; xx < N - 1; xx++) { const size_t r = ((size_t)z * N + yy) * N + xx; checked++; if (h1[r] != h2[r]) bad++; } \
        printf("march TY=%2d D=%d xcd=%d zseg=%3d: %.4f ms  %.3f of 8 TB/s  (interior sample: %zu of %zu differ)\n", 4 * LPW, D, (int)XCD, ZSEG, ms, bytes / ms / 1e6 / 8000, bad, checked); \
        fflush(stdout); \
    } while (0)
    double *vals; CK(hipMalloc(&vals, 8 * n * 7 + 64));
    { std::vector<double> hv(n); for (int u = 0; u < 7; u++) { for (size_t i = 0; i < n; i++) hv[i] = (u == 3 ? 6.0 : -1.0); CK(hipMemcpy(vals + (size_t)u * n, hv.data(), 8 * n, hipMemcpyHostToDevice)); } }
    const double bytesv = 73.0 * n;
#define RUNV(LPW, D, ZSEG) do { \
        const int tiles_x = N / 128, tiles_y = N / (4 * LPW), nseg = (N + ZSEG - 1) / ZSEG; \
        if (N % 128 || N % (4 * LPW)) break; \
        CK(hipMemset(y, 0, 8 * n)); \
        ms = timeit([&] { marchv_kernel<LPW, D, true, D><<<tiles_x * tiles_y * nseg, 256>>>(pat, x, y, N, ZSEG, vals, tiles_x, tiles_y); }, 5, 20); \
        CK(hipGetLastError()); \
        CK(hipMemcpy(h1.data(), y, 8 * n, hipMemcpyDeviceToHost)); \
        size_t bad = 0, checked = 0; \
        for (int z = 1; z < N - 1; z += 7) for (int yy = 1; yy < N - 1; yy += 5) for (int xx = 1; xx < N - 1; xx++) { const size_t r = ((size_t)z * N + yy) * N + xx; checked++; if (h1[r] != h2[r]) bad++; } \
        printf("march, values streamed (SoA) TY=%2d D=%d zseg=%3d: %.4f ms  %.3f of 8 TB/s on 73 B/row  (interior sample: %zu of %zu differ)\n", 4 * LPW, D, ZSEG, ms, bytesv / ms / 1e6 / 8000, bad, checked); \
        fflush(stdout); \
    } while (0)
    RUNV(1, 2, 171); RUNV(1, 2, 128); RUNV(1, 2, 64); RUNV(2, 2, 171); RUNV(2, 2, 128); RUNV(1, 3, 171);
    RUN(2, 2, true, 64); RUN(2, 3, true, 64); RUN(2, 4, true, 64);
    RUN(2, 3, true, 32); RUN(2, 3, true, 128); RUN(2, 3, false, 64);
    RUN(1, 3, true, 64); RUN(1, 4, true, 64); RUN(1, 6, true, 64);
    RUN(4, 2, true, 64); RUN(4, 3, true, 64);
    return 0;
}
