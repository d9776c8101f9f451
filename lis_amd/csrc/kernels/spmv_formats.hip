// spmv_formats.hip -- ELL / DIA / JAD / BSR sparse matrix-vector products for gfx950, f64 / i32.
//
// One lane owns one output row and adds its terms strictly in the stored order starting from +0.0,
// which is the rounding sequence of the reference's loops (each `y[i] += a*x` there is one rounded
// multiply and one rounded add), so results are bit-identical to the reference at any thread count:
//   ELL  src/matvec/lis_matvec_ell.c:113-128      DIA  src/matvec/lis_matvec_dia.c:148-172
//   JAD  src/matvec/lis_matvec_jad.c:170-196      BSR  src/matvec/lis_matvec_bsr.c:120-148 (+ RxC :152-858)
// Column-major ELL/DIA/JAD storage makes the per-lane streams perfectly coalesced across a wavefront.
#include "common.hpp"
#include "liship.h"

namespace {

constexpr int BLOCK = 256;

// XCD strips for the native ELL / DIA kernels (round 6; csr_kernels.hpp xcd_strip_unit is the same permutation): workgroup w runs on XCD w % 8.  With `plane`
// workgroups per grid plane (a multiple of 8) every XCD takes one eighth of every plane and walks the planes in order, so the +-plane neighbours of its rows are
// rows it multiplied itself two strips earlier: x crosses the fabric once instead of once per XCD that touches it.  A permutation of the launch's workgroups:
// rows are independent, the bits cannot depend on it (the fused dots publish their partial under the PERMUTED index, so their fold order is the natural one too).
int g_fmt_plane_rows = 0;           // liship_spmv_formats_set_plane: rows per plane for launches that are not told (0: natural order)
__device__ __forceinline__ int fmt_strip_unit(int w, int n, int plane)
{
    if (plane <= 0) return w;
    const int full = (n / plane) * plane;
    if (w >= full) return w;
    const int sb = plane >> 3, xcd = w & 7, slot = w >> 3;
    const int pl = slot / sb;
    return pl * plane + xcd * sb + (slot - pl * sb);
}

// ELL: lane owns ROWS consecutive rows (2 when n is even: 16 B value / 8 B index loads), UNROLL jagged columns
// in flight before the first use; column-major storage makes every load of a wavefront contiguous.
// lane partials of the fused reductions <w,y> (DOT >= 1) and <y,y> (DOT == 2) -> one value per workgroup
template <int DOT>
__device__ __forceinline__ void publish(double c0, double c1, double *__restrict__ partial, int stride, int slot = -1)
{
    __shared__ double scratch[BLOCK / WAVE];
    if (DOT == 0) return;
    if (slot < 0) slot = (int)blockIdx.x;
    const double t0 = block_sum<BLOCK / WAVE>(c0, scratch);
    if (threadIdx.x == 0) partial[slot] = t0;
    if (DOT == 2) {
        const double t1 = block_sum<BLOCK / WAVE>(c1, scratch);
        if (threadIdx.x == 0) partial[stride + slot] = t1;
    }
}

// CODED: one byte per column index (position of column - row in a sorted dictionary of <= 255 offsets, as for CSR:
// liship_ell_encode_indices), 9 B per slot instead of 12; same lanes, same rows, same order -> same bits.
template <int ROWS, int UNROLL, int DOT = 0, bool CODED = false>
__global__ __launch_bounds__(BLOCK)
void spmv_ell_kernel(int n, int maxnzr, const int *__restrict__ idx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y,
                     const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                     const double *__restrict__ guard = nullptr,
                     const unsigned char *__restrict__ codes = nullptr, const int *__restrict__ dict = nullptr,
                     int rb = 0, int re = -1,               // rows [rb, re) of the n (re < 0: all): a multi-rank job's interior / boundary parts
                     int xs_plane = 0)                      // XCD strips: workgroups per grid plane (fmt_strip_unit), 0 = natural order
{
    const int bid = fmt_strip_unit((int)blockIdx.x, (int)gridDim.x, xs_plane);
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    __shared__ int dictL[CODED ? 256 : 1];
    if (CODED) {
        for (int i = threadIdx.x; i < 256; i += BLOCK) dictL[i] = dict[i];
        __syncthreads();
    }
    const int r0 = rb + (bid * BLOCK + threadIdx.x) * ROWS;
    const bool active = r0 < (re < 0 ? n : re);
    if (!DOT && !active) return;
    const int r = active ? r0 : 0;                  // idle lanes of the last workgroup shadow row 0 (fused form: they
    double acc[ROWS];                               // must reach the workgroup reduction)
#pragma unroll
    for (int i = 0; i < ROWS; i++) acc[i] = 0.0;
    for (int j0 = 0; j0 < maxnzr; j0 += UNROLL) {
        double v[UNROLL][ROWS], xv[UNROLL][ROWS];
        int c[UNROLL][ROWS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            // slots past the last jagged column (7 columns in batches of 8: the eighth) load NOTHING -- the test is wave-uniform, a scalar branch.  They used to repeat
            // the last column's loads under a mask: one load in eight for nothing (round 6: 7-point ELL at 512^3 2.35 -> 2.2x ms)
            if (j0 + u < maxnzr) {
                const size_t k = (size_t)(j0 + u) * (size_t)n + (size_t)r;
                if (ROWS == 2) {
                    const v2f64 vv = load_stream(reinterpret_cast<const v2f64 *>(val + k));
                    v[u][0] = vv.x; v[u][ROWS - 1] = vv.y;
                    if (CODED) {
                        const unsigned short two = load_stream(reinterpret_cast<const unsigned short *>(codes + k));
                        c[u][0] = r + dictL[two & 255]; c[u][ROWS - 1] = r + 1 + dictL[two >> 8];
                    } else {
                        const v2i32 cc = load_stream(reinterpret_cast<const v2i32 *>(idx + k));
                        c[u][0] = cc.x; c[u][ROWS - 1] = cc.y;
                    }
                } else { v[u][0] = load_stream(val + k); c[u][0] = load_stream(idx + k); }
            } else {
#pragma unroll
                for (int i = 0; i < ROWS; i++) { v[u][i] = 0.0; c[u][i] = r; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int i = 0; i < ROWS; i++) xv[u][i] = (j0 + u < maxnzr) ? x[c[u][i]] : 0.0;
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                const double t = v[u][i] * xv[u][i];
                acc[i] += (j0 + u < maxnzr) ? t : 0.0;      // +0.0 leaves the sum's bits unchanged
            }
    }
    double c0 = 0.0, c1 = 0.0;
    if (active) {
        if (ROWS == 2) { v2f64 o; o.x = acc[0]; o.y = acc[ROWS - 1]; store_stream(reinterpret_cast<v2f64 *>(y + r), o); }
        else store_stream(y + r, acc[0]);
        if (DOT) {
#pragma unroll
            for (int i = 0; i < ROWS; i++) { c0 += wdot[r + i] * acc[i]; if (DOT == 2) c1 += acc[i] * acc[i]; }
        }
    }
    publish<DOT>(c0, c1, partial, gridDim.x, bid);
}

// DIA: same shape; a diagonal's offset is wave-uniform (scalar load), x[r + off] is contiguous across the wavefront.
// The reference skips the part of a diagonal that falls outside the matrix (lis_matvec_dia.c:152-158): masked here.
template <int ROWS, int UNROLL, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_dia_kernel(int n, int ncols, int nnd, const int *__restrict__ off,
                     const double *__restrict__ val, const double *__restrict__ x,
                     double *__restrict__ y, const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                     const double *__restrict__ guard = nullptr, int rb = 0, int re = -1, int xs_plane = 0)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int bid = fmt_strip_unit((int)blockIdx.x, (int)gridDim.x, xs_plane);
    const int r0 = rb + (bid * BLOCK + threadIdx.x) * ROWS;
    const bool active = r0 < (re < 0 ? n : re);
    if (!DOT && !active) return;
    const int r = active ? r0 : 0;
    double acc[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; i++) acc[i] = 0.0;
    for (int d0 = 0; d0 < nnd; d0 += UNROLL) {
        double v[UNROLL][ROWS], xv[UNROLL][ROWS];
        bool ok[UNROLL][ROWS];
        int o_[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) o_[u] = off[min(d0 + u, nnd - 1)];       // the batch's offsets first (scalar loads, one wait for all of them): taken one by one inside the
                                                                                   // slots below, every diagonal's x loads waited for a scalar-cache round trip of their own
#pragma unroll
        for (int u = 0; u < UNROLL; u++) asm volatile("" : "+s"(o_[u]));          // (the compiler sinks the loads back into the slots otherwise)
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (d0 + u < nnd) {                       // wave-uniform: slots past the last diagonal load nothing (they used to repeat the last diagonal's loads under a mask)
                const int d = d0 + u;
                const size_t k = (size_t)d * (size_t)n + (size_t)r;
                if (ROWS == 2) {
                    const v2f64 vv = load_stream(reinterpret_cast<const v2f64 *>(val + k));
                    v[u][0] = vv.x; v[u][ROWS - 1] = vv.y;
                } else v[u][0] = load_stream(val + k);
                const int o = o_[u];
#pragma unroll
                for (int i = 0; i < ROWS; i++) {
                    const int c = r + i + o;
                    ok[u][i] = c >= 0 && c < ncols;
                    xv[u][i] = x[ok[u][i] ? c : r];
                }
            } else {
#pragma unroll
                for (int i = 0; i < ROWS; i++) { v[u][i] = 0.0; xv[u][i] = 0.0; ok[u][i] = false; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                const double t = v[u][i] * xv[u][i];
                acc[i] += ok[u][i] ? t : 0.0;
            }
    }
    double c0 = 0.0, c1 = 0.0;
    if (active) {
        if (ROWS == 2) { v2f64 o; o.x = acc[0]; o.y = acc[ROWS - 1]; store_stream(reinterpret_cast<v2f64 *>(y + r), o); }
        else store_stream(y + r, acc[0]);
        if (DOT) {
#pragma unroll
            for (int i = 0; i < ROWS; i++) { c0 += wdot[r + i] * acc[i]; if (DOT == 2) c1 += acc[i] * acc[i]; }
        }
    }
    publish<DOT>(c0, c1, partial, gridDim.x, bid);
}

// JAD: a lane owns TWO adjacent slots of the length-sorted order (their entries are adjacent in every jagged
// diagonal: one 16 B value load + one 8 B index load when the diagonal starts on an even element), UNROLL jagged
// diagonals in flight.  Diagonals only get shorter, so a slot past the end of diagonal j is past the end of all
// later ones (masked, not branched).  y leaves through nt stores (perm is the identity on runs of equal-length rows).
template <int UNROLL, bool VEC>
__global__ __launch_bounds__(BLOCK)
void spmv_jad_kernel(int n, int maxnzr, const int *__restrict__ perm, const int *__restrict__ ptr,
                     const int *__restrict__ idx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y)
{
    const int s = (blockIdx.x * BLOCK + threadIdx.x) * 2;   // first of the lane's two slots
    if (s >= n) return;
    const bool two = s + 1 < n;
    const int row0 = perm[s], row1 = two ? perm[s + 1] : 0;
    double acc0 = 0.0, acc1 = 0.0;
    for (int j0 = 0; j0 < maxnzr; j0 += UNROLL) {
        double v0[UNROLL], v1[UNROLL], x0[UNROLL], x1[UNROLL];
        int c0[UNROLL], c1[UNROLL];
        bool ok0[UNROLL], ok1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int j = min(j0 + u, maxnzr - 1);
            const int b = ptr[j], len = ptr[j + 1] - b;   // wave-uniform
            const bool live = j0 + u < maxnzr;
            ok0[u] = live && s < len;
            ok1[u] = live && two && s + 1 < len;
            const int k = ok0[u] ? b + s : 0;
            if (VEC && ok1[u] && (k & 1) == 0) {                // both slots, 16 B aligned (k even <=> b even: wave-uniform but for the tail)
                const v2f64 vv = load_stream(reinterpret_cast<const v2f64 *>(val + k));
                const v2i32 cc = load_stream(reinterpret_cast<const v2i32 *>(idx + k));
                v0[u] = vv.x; v1[u] = vv.y; c0[u] = cc.x; c1[u] = cc.y;
            } else {
                v0[u] = load_stream(val + k); c0[u] = load_stream(idx + k);
                const int k1 = ok1[u] ? k + 1 : k;
                v1[u] = load_stream(val + k1); c1[u] = load_stream(idx + k1);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { x0[u] = x[c0[u]]; x1[u] = x[c1[u]]; }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const double t0 = v0[u] * x0[u], t1 = v1[u] * x1[u];
            acc0 += ok0[u] ? t0 : 0.0;                    // +0.0 terms leave the sum bit-unchanged
            acc1 += ok1[u] ? t1 : 0.0;
        }
        if (!ok0[UNROLL - 1]) break;
    }
    store_stream(y + row0, acc0);
    if (two) store_stream(y + row1, acc1);
}

// one lane per scalar row: blocks of the block row in stored order, block columns ascending --
// the accumulation order of both the generic and the unrolled RxC reference kernels
__global__ __launch_bounds__(BLOCK)
void spmv_bsr_kernel(int nrows, int bnr, int bnc, const int *__restrict__ bptr,
                     const int *__restrict__ bidx, const double *__restrict__ val,
                     const double *__restrict__ x, double *__restrict__ y)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= nrows) return;
    const int br = r / bnr, ii = r - br * bnr, bs = bnr * bnc;
    double acc = 0.0;
    for (int b = bptr[br]; b < bptr[br + 1]; b++) {
        const double *blk = val + (size_t)b * (size_t)bs + ii;
        const double *xb = x + (size_t)bidx[b] * (size_t)bnc;
        for (int j = 0; j < bnc; j++) acc += blk[(size_t)j * bnr] * xb[j];
    }
    y[r] = acc;
}

// BSR, square blocks up to 4x4: a workgroup owns BLOCK/BNR block rows.  Phase 1: lane = one block COLUMN
// (BNR contiguous values -> fully coalesced 8*BNR B loads, one x value), the BNR rounded products go to LDS.
// Phase 2: lane = one scalar row, sums its products block by block, column by column -- the order of
// lis_matvec_bsr.c:120-148 -- with the running sum kept in a register across LDS passes for long block rows.
template <int BNR, int BNC>
__global__ __launch_bounds__(BLOCK)
void spmv_bsr_tile_kernel(int nr, const int *__restrict__ bptr, const int *__restrict__ bidx,
                          const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int BS = BNR * BNC;
    constexpr int BRW = BLOCK / BNR;                 // block rows per workgroup
    constexpr int CHUNK = 4096 / BS;                 // blocks per LDS pass (32 KB of products)
    __shared__ __attribute__((aligned(16))) double prod[CHUNK * BS];
    const int br0 = blockIdx.x * BRW;
    const int br1 = min(br0 + BRW, nr);
    const int bb = bptr[br0], be = bptr[br1];
    const int L = threadIdx.x;
    const int mybr = br0 + L / BNR, myi = L % BNR;
    const bool rowlane = (L < BRW * BNR) && mybr < br1;
    int rs = 0, re = 0;
    if (rowlane) { rs = bptr[mybr]; re = bptr[mybr + 1]; }
    double acc = 0.0;
    // a lane takes UB block columns of a pass (a whole pass: UB * BLOCK >= CHUNK * BNC).  Their block indices are fetched one pass
    // ahead, so that a pass is ONE round trip (values and x together) instead of two (index, then x)
    constexpr int UB = (CHUNK * BNC + BLOCK - 1) / BLOCK;
    static_assert(UB <= 16, "block columns in flight per lane");
    int bi[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) {
        const int t = min(L + u * BLOCK, min(CHUNK, be - bb) * BNC - 1);
        bi[u] = bb < be ? bidx[bb + t / BNC] : 0;
    }
    for (int cb = bb; cb < be; cb += CHUNK) {
        const int nblk = min(CHUNK, be - cb);
        {
            double v[UB][BNR], xj[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int t = min(L + u * BLOCK, nblk * BNC - 1);      // clamped: the tail repeats the last column
                const int b = cb + t / BNC, j = t % BNC;
                const double *src = val + (size_t)b * BS + (size_t)j * BNR;
                if (BNR == 2 || BNR == 4) {
#pragma unroll
                    for (int i = 0; i < BNR; i += 2) {
                        const v2f64 vv = load_stream(reinterpret_cast<const v2f64 *>(src + i));
                        v[u][i] = vv.x; v[u][i + 1 < BNR ? i + 1 : i] = vv.y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < BNR; i++) v[u][i] = load_stream(src + i);
                }
                xj[u] = x[(size_t)bi[u] * BNC + j];
            }
            const int nb2 = min(CHUNK, be - (cb + CHUNK));              // the next pass's block indices (none after the last)
            if (nb2 > 0) {
#pragma unroll
                for (int u = 0; u < UB; u++) bi[u] = bidx[cb + CHUNK + min(L + u * BLOCK, nb2 * BNC - 1) / BNC];
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int t = L + u * BLOCK;
                if (t < nblk * BNC) {
                    double *dst = prod + (size_t)t * BNR;
                    if (BNR == 2 || BNR == 4) {          // 16 B stores: half the LDS instructions and bank conflicts
#pragma unroll
                        for (int i = 0; i < BNR; i += 2) {
                            v2f64 pr; pr.x = v[u][i] * xj[u]; pr.y = v[u][i + 1 < BNR ? i + 1 : i] * xj[u];
                            *reinterpret_cast<v2f64 *>(dst + i) = pr;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < BNR; i++) dst[i] = v[u][i] * xj[u];
                    }
                }
            }
        }
        __syncthreads();
        if (rowlane) {
            const int s = max(rs, cb), e = min(re, cb + nblk);
            constexpr int UP = 4;                        // blocks whose products are read before they are added (in order)
            for (int b = s; b < e; b += UP) {
                double pv[UP][BNC];
#pragma unroll
                for (int u = 0; u < UP; u++) {
                    const double *pp = prod + (size_t)(min(b + u, e - 1) - cb) * BS + myi;
#pragma unroll
                    for (int j = 0; j < BNC; j++) pv[u][j] = pp[j * BNR];
                }
#pragma unroll
                for (int u = 0; u < UP; u++) {
                    const bool ok = b + u < e;           // acc starts at +0.0: a +0.0 term leaves it bit-unchanged
#pragma unroll
                    for (int j = 0; j < BNC; j++) acc += ok ? pv[u][j] : 0.0;
                }
            }
        }
        __syncthreads();
    }
    if (rowlane) store_stream(y + (size_t)mybr * BNR + myi, acc);
}

// BSR 2x2 (Lis's default block size): phase 1 with a lane per BLOCK -- two 16 B value loads, ONE 16 B gather of
// the x pair, the four rounded products to LDS as two 16 B stores; phase 2 as above (lane = scalar row, products
// added block by block, column by column: lis_matvec_bsr.c:120-148).
__global__ __launch_bounds__(BLOCK)
void spmv_bsr22_kernel(int nr, const int *__restrict__ bptr, const int *__restrict__ bidx,
                       const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int BRW = BLOCK / 2;                   // block rows per workgroup
    constexpr int CHUNK = 1024;                      // blocks per LDS pass (32 KB of products)
    __shared__ __attribute__((aligned(16))) double prod[CHUNK * 4];
    const int br0 = blockIdx.x * BRW;
    const int br1 = min(br0 + BRW, nr);
    const int bb = bptr[br0], be = bptr[br1];
    const int L = threadIdx.x;
    const int mybr = br0 + (L >> 1), myi = L & 1;
    const bool rowlane = mybr < br1;
    int rs = 0, re = 0;
    if (rowlane) { rs = bptr[mybr]; re = bptr[mybr + 1]; }
    double acc = 0.0;
    constexpr int UB = CHUNK / BLOCK;                // blocks per lane and pass: the whole pass in flight, its indices fetched a pass ahead
    int c[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) c[u] = bb < be ? bidx[bb + min(L + u * BLOCK, min(CHUNK, be - bb) - 1)] : 0;
    for (int cb = bb; cb < be; cb += CHUNK) {
        const int nblk = min(CHUNK, be - cb);
        {
            v2f64 a0[UB], a1[UB], xv[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int b = cb + min(L + u * BLOCK, nblk - 1);        // clamped: the tail repeats the last block
                const v2f64 *src = reinterpret_cast<const v2f64 *>(val + (size_t)b * 4);
                a0[u] = load_stream(src);            // column 0: a00 a10
                a1[u] = load_stream(src + 1);        // column 1: a01 a11
                xv[u] = *reinterpret_cast<const v2f64 *>(x + (size_t)c[u] * 2);
            }
            const int nb2 = min(CHUNK, be - (cb + CHUNK));              // the next pass's block indices (none after the last)
            if (nb2 > 0) {
#pragma unroll
                for (int u = 0; u < UB; u++) c[u] = bidx[cb + CHUNK + min(L + u * BLOCK, nb2 - 1)];
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int t = L + u * BLOCK;
                if (t < nblk) {
                    v2f64 p0, p1;
                    p0.x = a0[u].x * xv[u].x; p0.y = a0[u].y * xv[u].x;
                    p1.x = a1[u].x * xv[u].y; p1.y = a1[u].y * xv[u].y;
                    reinterpret_cast<v2f64 *>(prod)[2 * t] = p0;
                    reinterpret_cast<v2f64 *>(prod)[2 * t + 1] = p1;
                }
            }
        }
        __syncthreads();
        if (rowlane) {
            const int s = max(rs, cb), e = min(re, cb + nblk);
            constexpr int UP = 4;                    // blocks whose products are read before they are added (in order)
            for (int b = s; b < e; b += UP) {
                double pv[UP][2];
#pragma unroll
                for (int u = 0; u < UP; u++) {
                    const double *pp = prod + (size_t)(min(b + u, e - 1) - cb) * 4 + myi;
                    pv[u][0] = pp[0]; pv[u][1] = pp[2];
                }
#pragma unroll
                for (int u = 0; u < UP; u++) {
                    const bool ok = b + u < e;       // acc starts at +0.0: a +0.0 term leaves it bit-unchanged
                    acc += ok ? pv[u][0] : 0.0;
                    acc += ok ? pv[u][1] : 0.0;
                }
            }
        }
        __syncthreads();
    }
    if (rowlane) store_stream(y + (size_t)mybr * 2 + myi, acc);
}

// Square BSR blocks (2x2 is Lis's default, 3x3 / 4x4 the usual FEM choices) the way the CSR row-gather kernel works:
// one wavefront per workgroup owns 64 block rows, stages the raw value (8 BS^2 B) and index (4 B) slices of their
// blocks in LDS with direct global->LDS loads (every instruction a fully coalesced 1 KiB), then a lane walks ITS
// block row in stored order: per block BS^2 values from LDS, ONE gather of the BS x values and the multiply-adds of
// lis_matvec_bsr.c:120-148 in their order (column by column inside a block: t_i += a_ij x_j for i = 0..BS-1, then
// the next j).  All BS rows of a block row live in one lane, y leaves as 16 B nt stores where BS is even.
// Block rows longer than the stage are walked in passes with the running sums kept in registers.
// One wavefront per workgroup and a stage of 18-25 KB leave 1.6 wavefronts per SIMD (profiles/r02_bsr_kernel_pmc.txt): nobody hides
// a lane's latencies, so a lane issues the x gathers of a whole batch of U blocks (its whole row, typically) before it touches the
// values -- one gather round trip per pass instead of one per 2-3 blocks: 3x3 0.941 -> 0.741 ms, 4x4 0.748 -> 0.664, 2x2 0.397 -> 0.368
// at 256^3 (65.8 / 69.1 / 75.8 % -> 83.5 / 77.9 / 81.8 % of 8 TB/s on the stored bytes).
constexpr int BSR_LANES = 64;
template <int BS, int CAP, int U, int DOT = 0>
__global__ __launch_bounds__(BSR_LANES)
void spmv_bsr_rows_kernel(int nr, const int *__restrict__ bptr, const int *__restrict__ bidx,
                          const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
                          int n = 0, const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                          const double *__restrict__ guard = nullptr)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    constexpr int BB = BS * BS;
    // even BS: one 16 B unit of padding after every 1 KiB (= one DMA instruction) of the value stage.  Block rows of one length put
    // the lanes a fixed stride apart (7 blocks of 128 B: lanes 0, 2, 4 ... in the same banks -- two thirds of the 4x4 kernel's
    // LDS cycles were bank conflicts, profiles/r02_bsr_kernel_pmc.txt); the skew spreads them.  A block never straddles a KiB.
    constexpr int UNITS = ((CAP + 4) * BB + 2 * WAVE + 1) / 2;
    __shared__ __attribute__((aligned(16))) double valL[2 * (UNITS + (BB % 2 == 0 ? UNITS / 64 + 1 : 0))];
    __shared__ __attribute__((aligned(16))) int idxL[CAP + 4 + 4 * WAVE];
    const int lane = threadIdx.x;
    const int br0 = blockIdx.x * BSR_LANES, br1 = min(br0 + BSR_LANES, nr);
    const int bb = bptr[br0], be = bptr[br1], nblk_total = bptr[nr];
    const int mybr = br0 + lane;
    const bool live = mybr < br1;
    int rs = 0, re = 0;
    if (live) { rs = bptr[mybr]; re = bptr[mybr + 1]; }
    double t[BS], wv[BS];
#pragma unroll
    for (int i = 0; i < BS; i++) {
        t[i] = 0.0;
        const int r = mybr * BS + i;                     // w is fetched before the blocks: its latency hides behind them
        wv[i] = (DOT != 0 && live && r < n) ? wdot[r] : 0.0;
    }
    for (int cb = bb; cb < be; cb += CAP) {
        const int ka = cb & ~3;                          // 16 B aligned start of both slices (4 | ka: 32 BS^2 B and 16 B)
        const int cend = min(cb + CAP, be);
        const int cnt = cend - ka;                       // staged blocks [ka, cend)
        const int np = (cnt * BB + 1) >> 1;              // 16 B pieces of the value slice (ka * BB is even)
        // odd BS: the last piece of the array's last block would read 8 B past its end -- that value goes alone
        const bool tail = (long long)ka * BB + 2LL * np > (long long)nblk_total * BB;
        const int npd = tail ? (cnt * BB) >> 1 : np;
        for (int p0 = 0; p0 < npd; p0 += WAVE) {
            const int p = min(p0 + lane, npd - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + (size_t)ka * BB) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0 + (BB % 2 == 0 ? p0 >> 6 : 0)), 16, 0, 2);
        }
        if (tail && lane == 0) valL[cnt * BB - 1] = val[(size_t)(ka + cnt) * BB - 1];
        const int nq = (cnt + 3) >> 2;
        if (ka + 4 * nq <= nblk_total) {
            for (int q0 = 0; q0 < nq; q0 += WAVE) {
                const int q = min(q0 + lane, nq - 1);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v4i32 *>(bidx + ka) + q),
                    (__attribute__((address_space(3))) void *)(reinterpret_cast<v4i32 *>(idxL) + q0), 16, 0, 2);
            }
        } else {                                         // the last few indices of the array: no 16 B reads past its end
            for (int i = lane; i < cnt; i += WAVE) idxL[i] = bidx[ka + i];
        }
        __syncthreads();
        const int s = max(rs, cb), e = min(re, cend);
        for (int b0 = s; b0 < e; b0 += U) {
            // all x gathers of the batch first (a lane's walk is a chain of gather round trips otherwise: with one wavefront per
            // workgroup and the stage bounding the occupancy there is nobody to hide them), then block by block the values from LDS
            double xv[U][BS];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int o = min(b0 + u, e - 1) - ka;    // clamped: repeats the row's last block
                const double *xp = x + (size_t)idxL[o] * BS;
                if (BS % 2 == 0) {
#pragma unroll
                    for (int q = 0; q < BS / 2; q++) {
                        const v2f64 vv = reinterpret_cast<const v2f64 *>(xp)[q];
                        xv[u][2 * q] = vv.x; xv[u][2 * q + 1 < BS ? 2 * q + 1 : 2 * q] = vv.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < BS; q++) xv[u][q] = xp[q];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int o = min(b0 + u, e - 1) - ka;
                double a[BB];
                if (BB % 2 == 0) {
                    const int unit = o * (BB / 2), at = unit + (unit >> 6);      // the stage's skew: see the DMA above
#pragma unroll
                    for (int q = 0; q < BB / 2; q++) {
                        const v2f64 vv = reinterpret_cast<const v2f64 *>(valL)[at + q];
                        a[2 * q] = vv.x; a[2 * q + 1 < BB ? 2 * q + 1 : 2 * q] = vv.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < BB; q++) a[q] = valL[o * BB + q];
                }
                const bool ok = b0 + u < e;               // +0.0 terms leave the sums bit-unchanged
#pragma unroll
                for (int j = 0; j < BS; j++)
#pragma unroll
                    for (int i = 0; i < BS; i++) {
                        const double pr = a[i + j * BS] * xv[u][j];
                        t[i] += ok ? pr : 0.0;
                    }
            }
        }
        __syncthreads();
    }
    if (live) {
        double *yp = y + (size_t)mybr * BS;
        if (BS % 2 == 0) {
#pragma unroll
            for (int q = 0; q < BS / 2; q++) { v2f64 out; out.x = t[2 * q]; out.y = t[2 * q + 1 < BS ? 2 * q + 1 : 2 * q]; store_stream(reinterpret_cast<v2f64 *>(yp) + q, out); }
        } else {
#pragma unroll
            for (int i = 0; i < BS; i++) store_stream(yp + i, t[i]);
        }
    }
    if (DOT != 0) {                                      // <w,y> (and <y,y>): one partial per workgroup = per wavefront, no LDS
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int i = 0; i < BS; i++) {
            const bool in = live && mybr * BS + i < n;   // rows of the padding do not exist
            c0 += in ? wv[i] * t[i] : 0.0;
            if (DOT == 2) c1 += in ? t[i] * t[i] : 0.0;
        }
        c0 = wave_sum(c0);
        if (lane == 0) partial[blockIdx.x] = c0;
        if (DOT == 2) { c1 = wave_sum(c1); if (lane == 0) partial[gridDim.x + blockIdx.x] = c1; }
    }
}

// Square BSR blocks with LONG block rows (the dofs-per-node finite-element pattern: 27 blocks per block row for hexahedra): a TEAM of
// T lanes per block row, the four-lanes-per-row shape of spmv_csr_pattern_team_kernel (spmv_csr.hip).  A wavefront owns 64 / T
// neighbouring block rows and exactly their value slice (LDS-DMA, 16-18 KB in flight per wavefront; no barrier: one wavefront per
// workgroup); lane t of a team owns blocks SEG t .. SEG t + SEG - 1 of its block row: their SEG indices, the SEG x BS gathers of x --
// one batch, ahead of the slice -- and their BS x BS x SEG products, formed once and kept in registers.  The BS row sums of a block row
// stay ONE chain each, block by block, column by column (lis_matvec_bsr.c:120-148, the unrolled 3x3 order :340-343): lane 0 adds its
// products to 0.0, lane 1 takes the sums through the LDS crossbar (ds_bpermute: registers to registers) and adds its own, and so on
// for T rounds; absent blocks add +0.0 (the sums start at +0.0 and can never be -0.0).  lane = (64 / T) t + i, so that the lanes
// holding segment t of neighbouring block rows -- neighbouring block columns -- are neighbours.  A wavefront with a block row of more
// than T x SEG blocks walks its rows one scalar row per lane from global memory (rare: the kernel is chosen by the MEAN length).
// The two-phase tile kernels above measured 57-67 % of the roofline on this pattern: every pass of theirs is two barriers, and the
// products go through LDS.
template <int BS, int T, int SEG>
__global__ __launch_bounds__(WAVE)
void spmv_bsr_team_kernel(int nr, const int *__restrict__ bptr, const int *__restrict__ bidx,
                          const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int BB = BS * BS, RPW = WAVE / T, CAPB = RPW * T * SEG;      // block rows per wavefront; blocks the stage holds
    constexpr int PIECES = ((CAPB + 1) * BB + 1) / 2;                       // 16 B pieces of the largest slice
    // even BS: one 16 B unit of padding after every KiB (= one DMA instruction) of the stage, as in spmv_bsr_rows_kernel: block rows of one length
    // put the lanes of a segment a fixed stride apart (26 blocks of 128 B: every lane in the same banks); a block never straddles a KiB
    constexpr bool SKEW = BB % 2 == 0;
    __shared__ __attribute__((aligned(16))) double valL[2 * (PIECES + (SKEW ? PIECES / 64 + 1 : 0)) + 2 * WAVE];
    const int lane = (int)threadIdx.x, i = lane % RPW, t = lane / RPW;
    const int br0 = (int)blockIdx.x * RPW, br1 = min(br0 + RPW, nr);
    const int k0 = bptr[br0], k1 = bptr[br1], nblk_total = bptr[nr];       // (uniform: scalar loads)
    const int br = min(br0 + i, br1 - 1);
    const bool live = br0 + i < br1;
    const int rs = bptr[br], re = bptr[br + 1], len = re - rs;
    if (k1 == k0) {                                                        // nothing stored in these block rows
        if (live && t == T - 1) for (int q = 0; q < BS; q++) y[(size_t)br * BS + q] = 0.0;
        return;
    }
    if (__any(len > T * SEG)) {                                            // a block row the teams cannot hold: one scalar row per lane, from memory
        for (int r = br0 * BS + lane; r < br1 * BS; r += WAVE) {
            const int b = r / BS, ii = r - b * BS;
            double acc = 0.0;
            for (int k = bptr[b]; k < bptr[b + 1]; k++) {
                const double *blk = val + (size_t)k * BB + ii;
                const double *xb = x + (size_t)bidx[k] * BS;
                for (int j = 0; j < BS; j++) acc += blk[(size_t)j * BS] * xb[j];
            }
            y[r] = acc;
        }
        return;
    }
    // this lane's blocks (clamped to the row's last one: always an address of the row; masked below)
    int kb[SEG], bi[SEG];
#pragma unroll
    for (int u = 0; u < SEG; u++) {
        kb[u] = min(max(min(rs + SEG * t + u, re - 1), rs), nblk_total - 1);
        bi[u] = bidx[kb[u]];
    }
    double xv[SEG][BS];
#pragma unroll
    for (int u = 0; u < SEG; u++) {
        const double *xp = x + (size_t)bi[u] * BS;
#pragma unroll
        for (int j = 0; j < BS; j++) xv[u][j] = xp[j];
    }
    __builtin_amdgcn_sched_barrier(0);                  // (indices, gathers, then the slice: a wait for the indices must not be a wait for the slice)
    const int ka = k0 & ~1;                             // 16 B aligned start of the slice
    const int cnt = k1 - ka;
    int np = (cnt * BB + 1) >> 1;                       // 16 B pieces
    const bool tail = (long long)ka * BB + 2LL * np > (long long)nblk_total * BB;      // odd BS: the array's last value has no 16 B piece
    if (tail) np--;
#pragma unroll
    for (int it = 0; it < (PIECES + WAVE - 1) / WAVE; it++) {
        const int p0 = it * WAVE;
        if (p0 < np) {
            const int p = min(p0 + lane, np - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + (size_t)ka * BB) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0 + (SKEW ? it : 0)), 16, 0, 2);
        }
    }
    if (tail && lane == 0) valL[cnt * BB - 1] = val[(size_t)(ka + cnt) * BB - 1];      // (odd BS only: no skew)
    __builtin_amdgcn_s_waitcnt(0);                      // this wavefront's slice and gathers have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double pm[SEG][BS][BS];                             // [block][column][row]: the products, in the order they are added
#pragma unroll
    for (int u = 0; u < SEG; u++) {
        const int unit = (kb[u] - ka) * (BB / 2);        // (even BS) 16 B units in front of the block
        const double *a = SKEW ? valL + 2 * (unit + (unit >> 6)) : valL + (size_t)(kb[u] - ka) * BB;
        const bool ok = SEG * t + u < len;
#pragma unroll
        for (int j = 0; j < BS; j++)
#pragma unroll
            for (int q = 0; q < BS; q++) { const double pr = a[q + j * BS] * xv[u][j]; pm[u][j][q] = ok ? pr : 0.0; }
    }
    double acc[BS];
#pragma unroll
    for (int q = 0; q < BS; q++) acc[q] = 0.0;
    const int from = ((lane - RPW) & (WAVE - 1)) * 4;
#pragma unroll
    for (int k = 0; k < T; k++) {
        double in[BS];
#pragma unroll
        for (int q = 0; q < BS; q++)
            in[q] = k == 0 ? 0.0 : __hiloint2double(__builtin_amdgcn_ds_bpermute(from, __double2hiint(acc[q])),
                                                    __builtin_amdgcn_ds_bpermute(from, __double2loint(acc[q])));
#pragma unroll
        for (int u = 0; u < SEG; u++)
#pragma unroll
            for (int j = 0; j < BS; j++)
#pragma unroll
                for (int q = 0; q < BS; q++) in[q] += pm[u][j][q];
#pragma unroll
        for (int q = 0; q < BS; q++) acc[q] = t == k ? in[q] : acc[q];
    }
    if (live && t == T - 1) {
#pragma unroll
        for (int q = 0; q < BS; q++) store_stream(y + (size_t)br * BS + q, acc[q]);
    }
}

int g_bsr_team = 1;              // liship_spmv_bsr_set_team: 0 keeps long block rows on the two-phase tile kernels (A/B)

inline int grid_for(int n) { return (n + BLOCK - 1) / BLOCK; }
// workgroups per grid plane for a whole-matrix launch of `grid` workgroups of `rows_per_wg` rows each (fmt_strip_unit), 0: natural order
inline int fmt_plane(int rows_per_wg, int grid)
{
    if (g_fmt_plane_rows <= 0) return 0;
    const long long pb = (long long)g_fmt_plane_rows / rows_per_wg;
    if (pb * rows_per_wg != g_fmt_plane_rows || pb < 64 || (pb & 7) || pb * 4 > grid) return 0;      // whole workgroups per plane, eight strips, at least four planes
    return (int)pb;
}

} // namespace

// the plane of a structured grid from an ELL matrix's own arrays (as liship_csr_plan_scan_band finds it for CSR): the largest |column - row| over the slots with owned
// columns, when at least half of the rows reach it -- the +-plane neighbours of a 3-D stencil.  Two passes over index[] at upload time; 0 when there is none.
namespace {
__global__ void ell_band_max(int n, int maxnzr, const int *__restrict__ idx, int *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (r < n) for (int j = 0; j < maxnzr; j++) { const int c = idx[(size_t)j * n + r]; if (c >= 0 && c < n) m = max(m, abs(c - r)); }
    for (int s = WAVE / 2; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}
__global__ void ell_band_count(int n, int maxnzr, const int *__restrict__ idx, const int *__restrict__ band, unsigned long long *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, B = band[0];
    bool hit = false;
    if (r < n) for (int j = 0; j < maxnzr; j++) { const int c = idx[(size_t)j * n + r]; hit = hit || (c >= 0 && c < n && abs(c - r) == B); }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(hit);
    if ((threadIdx.x & (WAVE - 1)) == 0 && b) atomicAdd(out, (unsigned long long)__builtin_popcountll(b));
}
}
extern "C" int liship_ell_scan_band(int n, int maxnzr, const int *idx, int *plane_rows, void *stream)
{
    if (!plane_rows) return LISHIP_ERR_ARG;
    *plane_rows = 0;
    if (n <= 0 || maxnzr <= 0 || !idx) return 0;
    hipStream_t st = as_stream(stream);
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), st);
    const int grid = (n + BLOCK - 1) / BLOCK;
    if (e == hipSuccess) { ell_band_max<<<grid, BLOCK, 0, st>>>(n, maxnzr, idx, reinterpret_cast<int *>(d)); e = hipGetLastError(); }
    if (e == hipSuccess) { ell_band_count<<<grid, BLOCK, 0, st>>>(n, maxnzr, idx, reinterpret_cast<const int *>(d), d + 1); e = hipGetLastError(); }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    const int band = (int)(h[0] & 0xffffffffull);
    if (band > 0 && 2 * h[1] >= (unsigned long long)n) *plane_rows = band;
    return 0;
}

// rows per plane of the structured grid the NEXT whole-matrix ELL / DIA launches work on (0: none -- natural workgroup order): the host layer sets it from the matrix
// it is about to multiply (one driving thread per process, as the Lis API requires: lis_device.c)
extern "C" int liship_spmv_formats_set_plane(int rows) { g_fmt_plane_rows = rows > 0 ? rows : 0; return 0; }

extern "C" int liship_spmv_ell_f64(int n, int maxnzr, const int *idx, const double *val,
                                   const double *x, double *y, void *stream)
{
    if (n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    if (maxnzr == 0) { HIP_TRY(hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, as_stream(stream))); return 0; }
    if ((n & 1) == 0 && aligned16(val) && aligned16(y) && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0)
        spmv_ell_kernel<2, 8><<<grid_for(n / 2), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, idx, val, x, y, nullptr, nullptr, nullptr, nullptr, nullptr, 0, -1, fmt_plane(2 * BLOCK, grid_for(n / 2)));
    else
        spmv_ell_kernel<1, 8><<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, idx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_spmv_dia_f64(int n, int ncols, int nnd, const int *off, const double *val,
                                   const double *x, double *y, void *stream)
{
    if (n < 0 || nnd < 0 || ncols < n) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    if (nnd == 0) { HIP_TRY(hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, as_stream(stream))); return 0; }
    if ((n & 1) == 0 && aligned16(val) && aligned16(y))
        spmv_dia_kernel<2, 8><<<grid_for(n / 2), BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, off, val, x, y, nullptr, nullptr, nullptr, 0, -1, fmt_plane(2 * BLOCK, grid_for(n / 2)));
    else
        spmv_dia_kernel<1, 8><<<grid_for(n), BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, off, val, x, y);
    LAUNCH_CHECK();
    return 0;
}


// Rows [rb, re) only (the array layout is that of all n rows): a multi-rank product runs the rows that reference no ghost column while the
// halo is in flight and the boundary rows after it -- rows are independent, so the parts write the bits of the whole launch.
extern "C" int liship_spmv_ell_rows_f64(int n, int maxnzr, const int *idx, const unsigned char *codes, const int *dict, const double *val,
                                        const double *x, double *y, int rb, int re, void *stream)
{
    if (n < 0 || maxnzr < 0 || rb < 0 || re > n) return LISHIP_ERR_ARG;
    if (rb >= re) return 0;
    hipStream_t st = as_stream(stream);
    if (maxnzr == 0) { HIP_TRY(hipMemsetAsync(y + rb, 0, sizeof(double) * (size_t)(re - rb), st)); return 0; }
    const bool pairs = (n & 1) == 0 && (rb & 1) == 0 && ((re - rb) & 1) == 0 && aligned16(val) && aligned16(y);
    if (codes && dict && pairs && (reinterpret_cast<uintptr_t>(codes) & 1u) == 0)
        spmv_ell_kernel<2, 8, 0, true><<<grid_for((re - rb) / 2), BLOCK, 0, st>>>(n, maxnzr, nullptr, val, x, y, nullptr, nullptr, nullptr, codes, dict, rb, re);
    else if (!idx) return LISHIP_ERR_ARG;
    else if (pairs && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0)
        spmv_ell_kernel<2, 8><<<grid_for((re - rb) / 2), BLOCK, 0, st>>>(n, maxnzr, idx, val, x, y, nullptr, nullptr, nullptr, nullptr, nullptr, rb, re);
    else
        spmv_ell_kernel<1, 8><<<grid_for(re - rb), BLOCK, 0, st>>>(n, maxnzr, idx, val, x, y, nullptr, nullptr, nullptr, nullptr, nullptr, rb, re);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int liship_spmv_dia_rows_f64(int n, int ncols, int nnd, const int *off, const double *val,
                                        const double *x, double *y, int rb, int re, void *stream)
{
    if (n < 0 || nnd < 0 || ncols < n || rb < 0 || re > n) return LISHIP_ERR_ARG;
    if (rb >= re) return 0;
    hipStream_t st = as_stream(stream);
    if (nnd == 0) { HIP_TRY(hipMemsetAsync(y + rb, 0, sizeof(double) * (size_t)(re - rb), st)); return 0; }
    if ((n & 1) == 0 && (rb & 1) == 0 && ((re - rb) & 1) == 0 && aligned16(val) && aligned16(y))
        spmv_dia_kernel<2, 8><<<grid_for((re - rb) / 2), BLOCK, 0, st>>>(n, ncols, nnd, off, val, x, y, nullptr, nullptr, nullptr, rb, re);
    else
        spmv_dia_kernel<1, 8><<<grid_for(re - rb), BLOCK, 0, st>>>(n, ncols, nnd, off, val, x, y, nullptr, nullptr, nullptr, rb, re);
    LAUNCH_CHECK();
    return 0;
}

// ---- ELL index coding (see spmv_csr.hip "coded indices"): the offsets column - row of all n*maxnzr slots, padding
// slots included (they point at their own row: offset 0)
namespace {
constexpr int ELL_TABLE = 1024, ELL_EMPTY = -2147483647 - 1;
__global__ void ell_collect_offsets(int n, long long slots, const int *__restrict__ idx, int *__restrict__ table, int *__restrict__ count)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= slots || count[0] > 255) return;
    const int off = idx[k] - (int)(k % n);
    unsigned h = ((unsigned)off * 2654435761u) >> 22;
    for (int probe = 0; probe < ELL_TABLE; probe++) {
        const int v = table[h];
        if (v == off) return;
        if (v == ELL_EMPTY) {
            const int old = atomicCAS(&table[h], ELL_EMPTY, off);
            if (old == ELL_EMPTY) { atomicAdd(count, 1); return; }
            if (old == off) return;
        }
        h = (h + 1) & (ELL_TABLE - 1);
        if (count[0] > 255) return;
    }
}
__global__ void ell_encode(int n, long long slots, const int *__restrict__ idx, const int *__restrict__ dict, int ndict,
                           unsigned char *__restrict__ codes)
{
    __shared__ int d[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) d[i] = dict[i];
    __syncthreads();
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= slots) return;
    const int off = idx[k] - (int)(k % n);
    int lo = 0, hi = ndict - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (d[mid] < off) lo = mid + 1; else hi = mid; }
    codes[k] = (unsigned char)lo;
}
} // namespace

// *codes (n*maxnzr bytes) and *dict (256 ints) are allocated on the device when the matrix qualifies (<= 255 distinct
// offsets, even n, 16 B aligned arrays), else left NULL; the caller frees them with liship_free.  Setup-time.
extern "C" int liship_ell_encode_indices(int n, int maxnzr, const int *idx, unsigned char **codes, int **dict, int *ndict, void *stream)
{
    if (!codes || !dict || !ndict || n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    *codes = nullptr; *dict = nullptr; *ndict = 0;
    if (n == 0 || maxnzr == 0 || (n & 1) || !idx) return 0;
    hipStream_t st = as_stream(stream);
    const long long slots = (long long)n * maxnzr;
    int *table = nullptr, host[ELL_TABLE + 1];
    HIP_TRY(hipMalloc(&table, sizeof(host)));
    for (int i = 0; i < ELL_TABLE; i++) host[i] = ELL_EMPTY;
    host[ELL_TABLE] = 0;
    hipError_t e = hipMemcpyAsync(table, host, sizeof(host), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { ell_collect_offsets<<<(unsigned)((slots + 255) / 256), 256, 0, st>>>(n, slots, idx, table, table + ELL_TABLE); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(host, table, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(table);
    if (e != hipSuccess) return (int)e;
    if (host[ELL_TABLE] > 255) return 0;
    int d[256], nd = 0;
    for (int i = 0; i < ELL_TABLE; i++) if (host[i] != ELL_EMPTY && nd < 256) d[nd++] = host[i];
    if (nd == 0 || nd > 255) return 0;
    for (int i = 1; i < nd; i++) { const int v = d[i]; int j = i - 1; while (j >= 0 && d[j] > v) { d[j + 1] = d[j]; j--; } d[j + 1] = v; }
    for (int i = nd; i < 256; i++) d[i] = d[nd - 1];
    e = hipMalloc(dict, sizeof(d));
    if (e == hipSuccess) e = hipMalloc(codes, (size_t)slots + 16);
    if (e == hipSuccess) e = hipMemcpyAsync(*dict, d, sizeof(d), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { ell_encode<<<(unsigned)((slots + 255) / 256), 256, 0, st>>>(n, slots, idx, *dict, nd, *codes); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (*codes) (void)hipFree(*codes); if (*dict) (void)hipFree(*dict); *codes = nullptr; *dict = nullptr; return (int)e; }
    *ndict = nd;
    return 0;
}

// y = A x from the coded form; want_sumsq < 0: no reduction, else the epilogue of liship_spmv_ell_dot_f64
extern "C" int liship_spmv_ell_coded_f64(int n, int maxnzr, const unsigned char *codes, const int *dict, const double *val,
                                         const double *x, double *y, const double *w, int want_sumsq, double *result,
                                         void *work, void *stream)
{
    if (n <= 0 || maxnzr <= 0 || !codes || !dict) return LISHIP_ERR_ARG;
    if ((n & 1) || !aligned16(val) || !aligned16(y) || (reinterpret_cast<uintptr_t>(codes) & 1u)) return LISHIP_ERR_ARG;
    const int grid = grid_for(n / 2);
    hipStream_t st = as_stream(stream);
    if (want_sumsq < 0) {
        spmv_ell_kernel<2, 8, 0, true><<<grid, BLOCK, 0, st>>>(n, maxnzr, nullptr, val, x, y, nullptr, nullptr, nullptr, codes, dict, 0, -1, fmt_plane(2 * BLOCK, grid));
        LAUNCH_CHECK();
        return 0;
    }
    if (liship_internal_ref_chunks()) return LISHIP_ERR_ARG;      // reference-order sums: the plain product, then one ordered pass
    if (!w || !result || !work) return LISHIP_ERR_ARG;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    if ((size_t)grid > slots) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    if (want_sumsq) spmv_ell_kernel<2, 8, 2, true><<<grid, BLOCK, 0, st>>>(n, maxnzr, nullptr, val, x, y, w, partial, liship_internal_guard(), codes, dict, 0, -1, fmt_plane(2 * BLOCK, grid));
    else            spmv_ell_kernel<2, 8, 1, true><<<grid, BLOCK, 0, st>>>(n, maxnzr, nullptr, val, x, y, w, partial, liship_internal_guard(), codes, dict, 0, -1, fmt_plane(2 * BLOCK, grid));
    LAUNCH_CHECK();
    return liship_internal_fold(grid, want_sumsq ? 2 : 1, grid, partial, spare, result, stream);
}

// ELL / DIA products with the reduction epilogue of liship_spmv_csr_dot_f64 (same contract: LISHIP_ERR_ARG when the
// fused form cannot serve the call -- odd n, unaligned arrays -- and the caller then uses product + dot)
extern "C" int liship_spmv_ell_dot_f64(int n, int maxnzr, const int *idx, const double *val, const double *x, double *y,
                                       const double *w, int want_sumsq, double *result, void *work, void *stream)
{
    if (n <= 0 || maxnzr <= 0 || !w || !result || !work || liship_internal_ref_chunks()) return LISHIP_ERR_ARG;
    if ((n & 1) || !aligned16(val) || !aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7u)) return LISHIP_ERR_ARG;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    const int grid = grid_for(n / 2);
    if ((size_t)grid > slots) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    if (want_sumsq) spmv_ell_kernel<2, 8, 2><<<grid, BLOCK, 0, as_stream(stream)>>>(n, maxnzr, idx, val, x, y, w, partial, liship_internal_guard(), nullptr, nullptr, 0, -1, fmt_plane(2 * BLOCK, grid));
    else            spmv_ell_kernel<2, 8, 1><<<grid, BLOCK, 0, as_stream(stream)>>>(n, maxnzr, idx, val, x, y, w, partial, liship_internal_guard(), nullptr, nullptr, 0, -1, fmt_plane(2 * BLOCK, grid));
    LAUNCH_CHECK();
    return liship_internal_fold(grid, want_sumsq ? 2 : 1, grid, partial, spare, result, stream);
}

extern "C" int liship_spmv_dia_dot_f64(int n, int ncols, int nnd, const int *off, const double *val, const double *x, double *y,
                                       const double *w, int want_sumsq, double *result, void *work, void *stream)
{
    if (n <= 0 || nnd <= 0 || ncols < n || !w || !result || !work || liship_internal_ref_chunks()) return LISHIP_ERR_ARG;
    if ((n & 1) || !aligned16(val) || !aligned16(y)) return LISHIP_ERR_ARG;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    const int grid = grid_for(n / 2);
    if ((size_t)grid > slots) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    if (want_sumsq) spmv_dia_kernel<2, 8, 2><<<grid, BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, off, val, x, y, w, partial, liship_internal_guard(), 0, -1, fmt_plane(2 * BLOCK, grid));
    else            spmv_dia_kernel<2, 8, 1><<<grid, BLOCK, 0, as_stream(stream)>>>(n, ncols, nnd, off, val, x, y, w, partial, liship_internal_guard(), 0, -1, fmt_plane(2 * BLOCK, grid));
    LAUNCH_CHECK();
    return liship_internal_fold(grid, want_sumsq ? 2 : 1, grid, partial, spare, result, stream);
}

extern "C" int liship_spmv_jad_f64(int n, int maxnzr, const int *perm, const int *ptr, const int *idx,
                                   const double *val, const double *x, double *y, void *stream)
{
    if (n < 0 || maxnzr < 0) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    if (maxnzr == 0) { HIP_TRY(hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, as_stream(stream))); return 0; }
    if (aligned16(val) && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0)
        spmv_jad_kernel<8, true><<<grid_for((n + 1) / 2), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, perm, ptr, idx, val, x, y);
    else
        spmv_jad_kernel<8, false><<<grid_for((n + 1) / 2), BLOCK, 0, as_stream(stream)>>>(n, maxnzr, perm, ptr, idx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}

// Block rows [brb, bre) only: a multi-rank product runs the block rows that reference no ghost block column while the halo is in flight and the boundary block
// rows after it.  A block-row range of a BSR matrix IS a BSR matrix -- bptr + brb still indexes the whole bindex / value arrays, y moves by brb * bnr -- so
// the kernels above serve it unchanged (the same kind of kernel as the whole matrix takes: the mean block-row length is the matrix's); rows are independent,
// so the parts write the bits of the whole launch.
extern "C" int liship_spmv_bsr_rows_f64(int nr, int bnnz, int bnr, int bnc, const int *bptr, const int *bidx, const double *val,
                                        const double *x, double *y, int brb, int bre, void *stream)
{
    if (nr < 0 || bnr < 1 || bnc < 1 || brb < 0 || bre > nr) return LISHIP_ERR_ARG;
    if (brb >= bre) return 0;
    const long long part = nr > 0 && bnnz >= 0 ? (long long)bnnz * (bre - brb) / nr : -1;
    return liship_spmv_bsr_nnz_f64(bre - brb, (int)part, bnr, bnc, bptr + brb, bidx, val, x, y + (size_t)brb * bnr, stream);
}

extern "C" int liship_spmv_bsr_set_team(int on) { g_bsr_team = on; return 0; }

extern "C" int liship_spmv_bsr_f64(int nr, int bnr, int bnc, const int *bptr, const int *bidx,
                                   const double *val, const double *x, double *y, void *stream)
{ return liship_spmv_bsr_nnz_f64(nr, -1, bnr, bnc, bptr, bidx, val, x, y, stream); }

// y = A x with result[0] = <w,y> (result[1] = <y,y> if want_sumsq) in the product's pass, for the square block sizes
// and block-row lengths the lane-per-block-row kernel serves; LISHIP_ERR_ARG otherwise (the caller then runs the
// product and the reduction separately) -- the contract of liship_spmv_csr_dot_f64
extern "C" int liship_spmv_bsr_dot_f64(int nr, int n, int bnnz, int bs, const int *bptr, const int *bidx, const double *val,
                                       const double *x, double *y, const double *w, int want_sumsq, double *result,
                                       void *work, void *stream)
{
    if (nr <= 0 || n <= 0 || bnnz < 0 || !w || !result || !work || liship_internal_ref_chunks()) return LISHIP_ERR_ARG;
    if (!aligned16(val) || !aligned16(x) || !aligned16(y) || !aligned16(bidx)) return LISHIP_ERR_ARG;
    const double mean = (double)bnnz / nr;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    const int grid = (nr + BSR_LANES - 1) / BSR_LANES;
    if ((size_t)grid > slots) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    hipStream_t st = as_stream(stream);
    const double *guard = liship_internal_guard();
#define GO(BS, CAP, U) do { if (want_sumsq) spmv_bsr_rows_kernel<BS, CAP, U, 2><<<grid, BSR_LANES, 0, st>>>(nr, bptr, bidx, val, x, y, n, w, partial, guard); \
                            else            spmv_bsr_rows_kernel<BS, CAP, U, 1><<<grid, BSR_LANES, 0, st>>>(nr, bptr, bidx, val, x, y, n, w, partial, guard); } while (0)
    if (bs == 2 && mean <= 12.0) GO(2, 512, 8);
    else if (bs == 3 && mean <= 16.0) GO(3, 352, 12);
    else if (bs == 4 && mean <= 12.0) GO(4, 160, 8);
    else return LISHIP_ERR_ARG;
#undef GO
    LAUNCH_CHECK();
    return liship_internal_fold(grid, want_sumsq ? 2 : 1, grid, partial, spare, result, stream);
}

// the same with the number of stored blocks known (the host layer knows it): block rows that are short on average
// take the lane-per-block-row kernel, long ones keep the lane-per-block(-column) kernels (as CSR's two kernels)
extern "C" int liship_spmv_bsr_nnz_f64(int nr, int bnnz, int bnr, int bnc, const int *bptr, const int *bidx,
                                       const double *val, const double *x, double *y, void *stream)
{
    if (nr < 0 || bnr < 1 || bnc < 1) return LISHIP_ERR_ARG;
    if (nr == 0) return 0;
    const long long rows = (long long)nr * bnr;
    if (rows > 0x7fffffffLL) return LISHIP_ERR_ARG;
    hipStream_t st = as_stream(stream);
    const double mean = bnnz >= 0 ? (double)bnnz / nr : 0.0;            // stored blocks per block row
    if (bnr == bnc && bnr <= 4 && aligned16(val)) {
        const int brw = BLOCK / bnr, grid = (nr + brw - 1) / brw;
        switch (bnr) {
        case 1: spmv_bsr_tile_kernel<1, 1><<<grid, BLOCK, 0, st>>>(nr, bptr, bidx, val, x, y); break;
        case 2:
            if (g_bsr_team && bnnz >= 0 && mean > 12.0) { spmv_bsr_team_kernel<2, 8, 4><<<(nr + 7) / 8, WAVE, 0, st>>>(nr, bptr, bidx, val, x, y); break; }
            if (aligned16(x) && aligned16(y) && aligned16(bidx) && (bnnz < 0 || mean <= 12.0))
                spmv_bsr_rows_kernel<2, 512, 8><<<(nr + BSR_LANES - 1) / BSR_LANES, BSR_LANES, 0, st>>>(nr, bptr, bidx, val, x, y);
            else if (aligned16(x)) spmv_bsr22_kernel<<<grid, BLOCK, 0, st>>>(nr, bptr, bidx, val, x, y);
            else              spmv_bsr_tile_kernel<2, 2><<<grid, BLOCK, 0, st>>>(nr, bptr, bidx, val, x, y);
            break;
        case 3:
            if (g_bsr_team && bnnz >= 0 && mean > 16.0) { spmv_bsr_team_kernel<3, 8, 4><<<(nr + 7) / 8, WAVE, 0, st>>>(nr, bptr, bidx, val, x, y); break; }
            if (aligned16(bidx) && bnnz >= 0 && mean <= 16.0) spmv_bsr_rows_kernel<3, 352, 12><<<(nr + BSR_LANES - 1) / BSR_LANES, BSR_LANES, 0, st>>>(nr, bptr, bidx, val, x, y);
            else                 spmv_bsr_tile_kernel<3, 3><<<grid, BLOCK, 0, st>>>(nr, bptr, bidx, val, x, y);
            break;
        default:
            // (4x4 long block rows stay with the tile kernel: a team's T rounds cost T x the additions, and at 16 per block that is what binds --
            //  <4, 16, 2> 62.6 %, <4, 8, 4> 62.0-63.8 % against 65.8-68.4 %, profiles/r03_bsr_team_kernel.txt; g_bsr_team = 2 selects it for A/B)
            if (g_bsr_team == 2 && bnnz >= 0 && mean > 12.0) { spmv_bsr_team_kernel<4, 8, 4><<<(nr + 7) / 8, WAVE, 0, st>>>(nr, bptr, bidx, val, x, y); break; }
            if (aligned16(x) && aligned16(y) && aligned16(bidx) && bnnz >= 0 && mean <= 12.0)
                spmv_bsr_rows_kernel<4, 160, 8><<<(nr + BSR_LANES - 1) / BSR_LANES, BSR_LANES, 0, st>>>(nr, bptr, bidx, val, x, y);
            else
                spmv_bsr_tile_kernel<4, 4><<<grid, BLOCK, 0, st>>>(nr, bptr, bidx, val, x, y);
            break;
        }
    } else
        spmv_bsr_kernel<<<grid_for((int)rows), BLOCK, 0, st>>>((int)rows, bnr, bnc, bptr, bidx, val, x, y);
    LAUNCH_CHECK();
    return 0;
}
