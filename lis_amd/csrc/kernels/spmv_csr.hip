// spmv_csr.hip -- CSR sparse matrix-vector product for gfx950 (MI355X), f64 values / i32 indices.
//
// Replaces the OpenMP loop of lis_matvec_csr (reference src/matvec/lis_matvec_csr.c:90-110).
// HBM-bound (12 B per non-zero + 20 B per row, 0.135 flop/B): no MFMA, everything below is about
// keeping ~6 TB/s of coalesced streams in flight and making the x[] gather cheap for the L1/TA.
//
//   * Row split balanced on merge-path coordinates.  Row block b owns the whole rows r whose coordinate
//     key(r) = r + ptr[r] falls in [b*WORK, (b+1)*WORK).  The split blk[b] = {row, ptr[row]} is built
//     once per matrix on the device (csr_plan_kernel: one binary search per block), costs 8 B per
//     ~1400 items, and one 16 B scalar load hands a workgroup its row range AND non-zero range.
//     Boundaries are pulled down to multiples of 16 rows when that is cheap, so every workgroup
//     writes whole 128 B lines of y.
//   * Default kernel ("row gather"): the raw value[]/index[] slice of the block goes global -> LDS
//     with 16 B-per-lane LDS-DMA (global_load_lds_dwordx4 nt: no VGPR round trip, no ds_write); then
//     lane r walks ITS row in LDS, gathers x[] itself and forms the row sum in registers, strictly
//     in stored order from +0.0 with one rounded multiply and one rounded add per term -- the
//     reference's rounding sequence, so y is bit-identical to lis_matvec_csr.
//     Consecutive lanes own consecutive rows: for banded / stencil matrices the j-th gather of a
//     wavefront touches 4-8 contiguous cache lines instead of ~20 when lanes own consecutive
//     non-zeros; the gather was the dominant cost once the streams ran at speed.
//   * y is written with non-temporal stores (a plain 8 B/row store stream next to the read streams
//     costs 20 % of the bandwidth on this chip, tools/ubench_write.hip).
//   * Fallback kernel ("products"): lanes own consecutive non-zeros, products value*x are parked in
//     LDS and summed per row in order.  Serves unaligned user arrays and rows longer than the LDS
//     stage (continued pass by pass by the lane that owns them, still in order).
// No __shfl tree is used for the row sums on purpose: a tree changes the association order and would
// break bit parity with the reference; the reductions that ARE trees live in vector_ops.hip.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "common.hpp"
#include "liship.h"

#include "csr_kernels.hpp"
#include "csr_plan.hpp"

// ------------------------------------------------------------------------------ launchers: which plan runs which kernel
namespace {

struct LaunchArgs {
    const int *ptr, *idx; const double *val, *x; double *y; const v2i32 *blk;
    int bfirst, nb, rb, re, nnz;
    hipStream_t st;
    const unsigned char *codes = nullptr;   // one-byte column codes + their dictionary, when the plan has them
    const int *dict = nullptr;
    const unsigned short *lcol = nullptr;   // block-local columns (positions, lists, list offsets), when the plan has them
    const int *dcol = nullptr, *doff = nullptr;
    double acc0 = 0.0;                      // what a row sum starts from (Rows)
    const unsigned char *rowpat = nullptr;  // row patterns (pattern per row, relative row starts, table), when the plan has them
    const unsigned short *rowrel = nullptr;
    const int *ptab = nullptr; int ptab_len = 0, npat1 = 0;
    const v4i32 *ptab8 = nullptr;
    const v4i32 *vrec = nullptr;            // value records (with ptab8), when the plan has them and they are switched on
    const int *order = nullptr;             // launch order of the products kernel (whole-matrix launches of a plan that has one)
    const v4i32 *vrecw = nullptr;           // wide value records (rows of up to 32 entries), when the plan has them and they are switched on
    const liship_csr_plan_s *plan = nullptr; // (set by the launchers) the dominant-pattern records live there
    const int *rowmap = nullptr;            // reordered plans: where row r of the (renumbered) matrix goes in y -- the block-local kernel only
};


// XCD strips of the 7-offset pattern kernel: whole-matrix launches of a plan that found a plane (xs_plane), unless switched off
// in units of `unit_rows` rows (the mean rows of a row block, the 64 rows of a team workgroup): units per plane, a multiple of 8, for launches over `units` units
inline int xcd_strip_plane(const liship_csr_plan_s *P, double unit_rows, long long units)
{
    if (!g_xcd_strips || !P || P->xs_rows <= 0 || unit_rows <= 0.0) return 0;
    const long long pb = (long long)((double)P->xs_rows / unit_rows + 0.5);
    if (pb < 64 || pb * 4 > units) return 0;                  // planes too small to matter / fewer than four of them
    return (int)(pb / 8) * 8;
}
inline int xcd_strips(const LaunchArgs &a)          // (row-range launches too -- the interior rows of a multi-rank slab: the strips are cut from the launch's own blocks)
{
    if (!a.plan || a.nb <= 0 || a.plan->nblocks <= 0) return 0;
    return xcd_strip_plane(a.plan, (double)a.plan->n / a.plan->nblocks, a.nb);
}


// the smallest offset distance (in rows) beyond the diagonal's neighbours that the dominant pattern has on BOTH sides: +-n of a 3-D stencil
static int dom_stride(const DomRec &D)
{
    int S = 0;
    for (int u = 0; u < 7; u++) {
        const int e = D.off[u] / 8;
        if (e > 1 && (S == 0 || e < S)) { bool both = false; for (int v = 0; v < 7; v++) both = both || D.off[v] == -8 * e; if (both) S = e; }
    }
    return S;
}

template <int G, int U>
void launch_rowgather(const LaunchArgs &a)
{
    constexpr Geometry g = kGeom[G];
    spmv_csr_rowgather_kernel<g.block, g.work, U>
        <<<a.nb, g.block, 0, a.st>>>(a.ptr, a.idx, a.val, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, xcd_strips(a), nullptr, nullptr, nullptr, 0, a.rowmap);
}

template <int G, int VEC>
void launch_products(const LaunchArgs &a)
{
    constexpr Geometry g = kGeom[G];
    spmv_csr_products_kernel<g.block, g.work, VEC>
        <<<a.nb, g.block, 0, a.st>>>(a.ptr, a.idx, a.val, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0},
                                     nullptr, nullptr, nullptr, 0, a.order);
}

static bool block_rows_serve(const liship_csr_plan_s *P, int rb, int re)      // the block-row kernel for this row range? (whole block rows only)
{
    return P && P->bd.len > 0 && P->bdrec && P->bstage && g_block_rows && g_team && rb % P->bd.b == 0 && re % P->bd.b == 0;
}

// the shape of the 27-point marching kernel for the rows [a.rb, a.re): whole planes of a plan whose grid is a box (try_box27).  Tiles of 128 columns x 8 lines (x 4 when
// eight does not divide the lines of a plane), segments of planes so that the launch has about three workgroups per CU (the 7-point kernel's choice).
static bool box27_shape(const LaunchArgs &a, Box27 &M, int &lpw)
{
    const liship_csr_plan_s *P = a.plan;
    if (!P || P->b27.S <= 0 || !g_dom_march || g_variant != 0 || !aligned16(a.y) || !g_row_values || a.acc0 != 0.0 || __builtin_signbit(a.acc0)) return false;
    M = P->b27;
    const int S = M.S, SO = M.SO;
    if (a.rb < 0 || a.re > P->n || a.rb % SO != 0 || a.re % SO != 0) return false;
    const int z0 = a.rb / SO, z1 = a.re / SO, planes = z1 - z0;
    if (planes < 1 || (planes < 8 && g_dom_march == 1)) return false;
    const int lines = SO / S;
    lpw = ((lines % 8 == 0 || lines % 4 != 0) && g_dom_march != 3) ? 2 : 1;      // (3: four lines per tile everywhere -- tests of that instantiation; lines % 4 == 0 keeps the whole tiles of four)
    const int tiles_x = (S + 127) / 128, tiles_y = (lines + 4 * lpw - 1) / (4 * lpw), tiles = tiles_x * tiles_y;      // (round 5: the last tile of a line / of a plane may be partial)
    if (g_dom_march == 1 && (S % 128 != 0 || lines % 4 != 0) && (long long)planes * SO < (4ll << 20)) return false;      // partial tiles pay on large grids only (the 7-point kernel's measurement)
    // (256^3, same box: three workgroups per CU 0.0591 ms, two 0.0613, four 0.0630, six 0.0633; tiles of four lines 0.0640 at best -- profiles/EXPERIMENTS.md)
    int nseg = (3 * 256 + tiles - 1) / tiles;
    if (nseg > planes / 8) nseg = planes / 8;
    if (nseg < 1) nseg = 1;
    const int zseg = (planes + nseg - 1) / nseg;
    nseg = (planes + zseg - 1) / zseg;
    M.tiles_x = tiles_x; M.tiles_y = tiles_y; M.zseg = zseg; M.nseg = nseg; M.z0 = z0; M.z1 = z1; M.wgs = tiles * nseg;
    if (M.wgs < 64 && g_dom_march == 1) return false;                // (a handful of workgroups walking a small grid: the staged kernel's thousands of independent wavefronts win)
    M.xcd = M.wgs >= 8 * NUM_XCD ? 1 : 0;
    return true;
}

// the same for the 2 x 2 block rows of a 7-point box grid (try_block2_march): tiles of 128 columns x 8 lines, whole planes; sums start at +0.0 only
static bool block2_shape(const LaunchArgs &a, Block2March &M)
{
    const liship_csr_plan_s *P = a.plan;
    if (!P || P->b2.S <= 0 || !g_dom_march || g_variant != 0 || !aligned16(a.y) || !g_row_values || a.acc0 != 0.0 || __builtin_signbit(a.acc0)) return false;
    M = P->b2;
    const int S = M.S, SO = M.SO;
    if (a.rb < 0 || a.re > P->n || a.rb % SO != 0 || a.re % SO != 0) return false;
    const int z0 = a.rb / SO, z1 = a.re / SO, planes = z1 - z0;
    if (planes < 1 || (planes < 8 && g_dom_march == 1)) return false;
    const int tiles_x = (S + 127) / 128, tiles_y = (SO / S + 7) / 8, tiles = tiles_x * tiles_y;      // (round 5: the last tile of a line / of a plane may be partial)
    if (g_dom_march == 1 && (S % 128 != 0 || (SO / S) % 8 != 0) && (long long)planes * SO < (4ll << 20)) return false;      // partial tiles pay on large grids only (the 7-point kernel's measurement)
    int nseg = (3 * 256 + tiles - 1) / tiles;
    if (nseg > planes / 8) nseg = planes / 8;
    if (nseg < 1) nseg = 1;
    const int zseg = (planes + nseg - 1) / nseg;
    nseg = (planes + zseg - 1) / zseg;
    M.tiles_x = tiles_x; M.tiles_y = tiles_y; M.zseg = zseg; M.nseg = nseg; M.z0 = z0; M.z1 = z1; M.wgs = tiles * nseg;
    if (M.wgs < 64 && g_dom_march == 1) return false;
    M.xcd = M.wgs >= 8 * NUM_XCD ? 1 : 0;
    return true;
}

// rows of up to 32 entries whose values ride in wide records: x staged per wavefront, the dominant pattern in scalar registers (variant 0x4000: the
// gathering kernel on plan row blocks, A/B)
static bool launch_wide(const LaunchArgs &a, const double *guard, int dot = 0, const double *w = nullptr, double *partial = nullptr, int pstride = 0, int *wgs_out = nullptr)
{
    const liship_csr_plan_s *P = a.plan;
    if (block_rows_serve(P, a.rb, a.re) && !(g_variant & 0x4000)) {      // one lane per block row (the row form of a b x b blocked stencil)
        const int b = P->bd.b, rows = a.re - a.rb, wgs = (rows + 256 * b - 1) / (256 * b);
        {   // 2 x 2 blocks of a 7-point box grid: whole planes march (spmv_csr_block2_march_kernel)
            Block2March M;
            if (rows > 0 && block2_shape(a, M)) {
                if (wgs_out) *wgs_out = M.wgs;
                const bool ws = dot != 0 && w != a.x;
                const bool pt = M.S % 128 != 0 || (M.SO / M.S) % 8 != 0;      // partial tiles: an instantiation of their own (the whole-tile one carries none of their tests)
#define GOB2(OD, DT, WS_) do { if (pt) spmv_csr_block2_march_kernel<OD, DT, WS_, true><<<M.wgs, 256, 0, a.st>>>(a.x, a.y, M, P->bd.maxcol + 1, w, partial, guard, pstride); \
                               else spmv_csr_block2_march_kernel<OD, DT, WS_, false><<<M.wgs, 256, 0, a.st>>>(a.x, a.y, M, P->bd.maxcol + 1, w, partial, guard, pstride); } while (0)
#define GOB2D(OD) do { if (dot == 0) GOB2(OD, 0, false); else if (dot == 1) { if (ws) GOB2(OD, 1, true); else GOB2(OD, 1, false); } else { if (ws) GOB2(OD, 2, true); else GOB2(OD, 2, false); } } while (0)
                if (M.ord == 0) GOB2D(0); else if (M.ord == 1) GOB2D(1); else GOB2D(2);
#undef GOB2D
#undef GOB2
                return true;
            }
        }
        if (wgs_out) *wgs_out = rows > 0 ? wgs : 0;
        if (rows <= 0) return true;
        const int xcap = (P->bd.slots + 1) & ~1, nl = (P->bd.slots + 2 * WAVE - 1) / (2 * WAVE);
#define GOB(B_, NL_, DT) spmv_csr_blockrows_staged_kernel<B_, NL_, DT><<<wgs, 256, sizeof(double) * 4 * (size_t)xcap, a.st>>>( \
        a.rowpat, P->bdrec, P->bstage, a.vrecw, a.npat1 - 1, a.x, a.y, Rows{a.rb, a.re, a.acc0}, P->bd, xcap, guard, w, partial, pstride, xcd_strip_plane(P, 256.0 * b, wgs))
#define GOBD(DT) do { if (b == 2) { if (nl <= 4) GOB(2, 4, DT); else GOB(2, 6, DT); } else if (b == 3) { if (nl <= 6) GOB(3, 6, DT); else GOB(3, 8, DT); } \
                      else { if (nl <= 8) GOB(4, 8, DT); else GOB(4, 12, DT); } } while (0)
        if (dot == 0) GOBD(0); else if (dot == 1) GOBD(1); else GOBD(2);
#undef GOBD
#undef GOB
        return true;
    }
    if (!P || !P->wdrec || !P->wstage || P->wd.len <= 0 || !g_team || (g_variant & 0x4000)) return false;
    {   // the 27-point box stencil: whole planes march (spmv_csr_box27_march_kernel)
        Box27 M;
        int lpw = 0;
        if (box27_shape(a, M, lpw)) {
            if (wgs_out) *wgs_out = M.wgs;
            const bool ws = dot != 0 && w != a.x;
            const bool pt = M.S % 128 != 0 || (M.SO / M.S) % (4 * lpw) != 0;      // partial tiles: an instantiation of their own (the whole-tile one carries none of their tests)
#define GO27(LPW_, DT, WS_) do { if (pt) spmv_csr_box27_march_kernel<LPW_, DT, WS_, true><<<M.wgs, 256, 0, a.st>>>(a.x, a.y, a.acc0, M, P->wd.maxcol + 1, w, partial, guard, pstride); \
                                 else spmv_csr_box27_march_kernel<LPW_, DT, WS_, false><<<M.wgs, 256, 0, a.st>>>(a.x, a.y, a.acc0, M, P->wd.maxcol + 1, w, partial, guard, pstride); } while (0)
#define GO27D(LPW_) do { if (dot == 0) GO27(LPW_, 0, false); else if (dot == 1) { if (ws) GO27(LPW_, 1, true); else GO27(LPW_, 1, false); } \
                         else { if (ws) GO27(LPW_, 2, true); else GO27(LPW_, 2, false); } } while (0)
            if (lpw == 2) GO27D(2); else GO27D(1);
#undef GO27D
#undef GO27
            return true;
        }
    }
    constexpr int CH = 1;                            // chunks of 64 rows per wavefront
    const int rows = a.re - a.rb, wgs = (rows + 256 * CH - 1) / (256 * CH);
    if (wgs_out) *wgs_out = rows > 0 ? wgs : 0;
    if (rows <= 0) return true;
    const int xcap = (P->wd.slots + 1) & ~1, nl = (P->wd.slots + 2 * WAVE - 1) / (2 * WAVE);
#define GOW(NL, DT) spmv_csr_valuerecw_staged_kernel<256, NL, CH, DT><<<wgs, 256, sizeof(double) * 4 * (size_t)xcap, a.st>>>( \
        a.rowpat, P->wdrec, P->wstage, a.vrecw, a.npat1 - 1, a.x, a.y, Rows{a.rb, a.re, a.acc0}, P->wd, xcap, guard, w, partial, pstride, xcd_strip_plane(P, 256.0 * CH, wgs))
#define GOWD(DT) do { if (nl <= 2) GOW(2, DT); else if (nl <= 3) GOW(3, DT); else if (nl <= 5) GOW(5, DT); else if (nl <= 6) GOW(6, DT); else GOW(8, DT); } while (0)
    if (dot == 0) GOWD(0); else if (dot == 1) GOWD(1); else GOWD(2);
#undef GOWD
#undef GOW
    return true;
}

// patterned rows of 8..32 entries, values streamed: four lanes per row, with x staged per wavefront when the plan found the dominant pattern's
// runs (variant 0x4000: the gathers of the first form, A/B)
static void launch_team(const LaunchArgs &a, const double *guard)
{
    const liship_csr_plan_s *P = a.plan;
    const int rows = a.re - a.rb, wgs = (rows + 63) / 64;
    if (rows <= 0) return;
    // (measured and dropped, profiles/EXPERIMENTS.md: XCD strips for these kernels -- neutral, they stage x per wavefront and are not bound by the fabric --; one lane
    //  per row with values and x staged -- 3 % faster on a repeated product, 3 % slower inside the Krylov loops)
    if (P->prec_slot && P->tr.nruns > 0 && !(g_variant & 0x4000)) {
        const int vcap = 16 * P->tr.maxlen + 48, xcap = (P->tr.slots + 1) & ~1;
#define GOT(NL) spmv_csr_pattern_team_staged_kernel<256, NL><<<wgs, 256, sizeof(double) * 4 * (size_t)(vcap + xcap), a.st>>>( \
            a.ptr, a.val, a.rowpat, P->prec36, P->prec_slot, a.x, a.y, Rows{a.rb, a.re, a.acc0}, a.nnz, P->tr, vcap, xcap, guard)
        if (P->tr.slots <= 2 * WAVE) GOT(1); else GOT(2);
#undef GOT
    } else
        spmv_csr_pattern_team_kernel<256><<<wgs, 256, 0, a.st>>>(a.ptr, a.val, a.rowpat, P->prec36, a.x, a.y, Rows{a.rb, a.re, a.acc0}, a.nnz, guard);
}

// the dominant-pattern product of a plan with value records (spmv_csr_valuerec_dom_kernel), plain or with the fused dots (a partial per workgroup: count_out)
// its shape for the rows [a.rb, a.re): the tiles, the run length of the XCD order; returns the number of workgroups that have rows (= partials of the fused form)
// the z-marching form (spmv_csr_valuerec_march_kernel) for the rows [a.rb, a.re)?  Whole planes of a grid whose dominant pattern is the 7-point stencil in ascending
// slot order, lines a multiple of 128 long, a multiple of 8 lines per plane.  Segments of planes: as long as possible while the launch has about three workgroups per CU.
static bool dom_march_shape(const LaunchArgs &a, DomMarch &M)
{
    const liship_csr_plan_s *P = a.plan;
    if (!g_dom_march || g_variant != 0 || !P || P->dom.mask != 0x7f || !aligned16(a.y) || P->dom_xlen < P->n) return false;
    const DomRec &D = P->dom;
    int S = 0, SO = 0, perm = 0;
    if (!dom_seven_point(D, S, SO, perm)) return false;
    const int order = perm == 0x1ac688 ? 0 : perm == 0x0e2a70 ? 1 : 2;      // (slot u at bits 3u: ascending 0,1,2,3,4,5,6; the generators' 0,6,1,5,2,4,3)
    // lines of any even length from 128 on (round 5: the last tile of a line is partial -- at least two pairs wide, so that its left and right halo columns have a lane each)
    // ... and planes of any number of lines from 8 on (the last tile of a plane holds fewer than its 8 lines); lines shorter than a tile from 64 columns on (half the lanes idle at worst)
    if (S < 64 || S % 2 != 0 || (S % 128 != 0 && S % 128 < 4) || SO <= S || SO % S != 0 || SO / S < 8) return false;
    if (a.rb % SO != 0 || a.re % SO != 0 || a.re > P->n || a.rb < 0) return false;
    const int z0 = a.rb / SO, z1 = a.re / SO, planes = z1 - z0;
    if (planes < 8) return false;
    const int tiles_x = (S + 127) / 128, tiles_y = (SO / S + 7) / 8, tiles = tiles_x * tiles_y;
    int nseg = (3 * 256 + tiles - 1) / tiles;                 // (512^3, 256 tiles: three segments 0.385 ms, four 0.398, two 0.37-0.42, one 0.52; tiles of 16 lines or three planes ahead: within the noise)
    if (nseg > planes / 8) nseg = planes / 8;
    if (nseg < 1) nseg = 1;
    const int zseg = (planes + nseg - 1) / nseg;
    nseg = (planes + zseg - 1) / zseg;
    M = DomMarch{S, SO, tiles_x, tiles_y, zseg, nseg, z0, z1, tiles * nseg, 0, perm, order, P->n / SO, P->box_modes};
    // the general form (patterns with values of their own, foreign patterns, padding terms: the waterfall in every plane) loses to the gathering kernel -- ELL / DIA row
    // forms at 256^3: 0.105 / 0.086 against 0.057 / 0.050 ms -- so by default only launches inside the box planes or of plans whose other patterns are plain masks march
    if (g_dom_march == 1 && !(P->dom_simple || (z0 >= P->box_z0 && z1 <= P->box_z1)) ) return false;
    if (g_dom_march == 1 && order == 2 && !P->dom_simple) return false;
    // partial tiles pay from ~4 M rows on (192^3: 0.031 -> 0.021 ms, 300^3: 0.115 -> 0.076; 150^3: 0.017 -> 0.019, 100^3: 0.0069 -> 0.0111, 64^3: 0.0046 -> 0.010 -- a few
    // hundred marching workgroups lose to the gathering kernel's thousands of independent wavefronts); grids of whole tiles keep the round-4 rule below
    if (g_dom_march == 1 && (S % 128 != 0 || (SO / S) % 8 != 0) && (S < 128 || (long long)planes * SO < (4ll << 20))) return false;
    if (M.wgs < 64 && g_dom_march == 1) return false;      // (a handful of workgroups walking a small grid: the gathering kernel's thousands of independent wavefronts win; 2 / 3: at any size, tests)
    M.xcd = M.wgs >= 8 * NUM_XCD ? 1 : 0;
    return true;
}

static long long dom_gather_shape(const LaunchArgs &a, DomTile &TL, int &run)
{
    const liship_csr_plan_s *P = a.plan;
    const long long rows = (long long)a.re - a.rb;
    // lane -> row mapping.  Default: TILES of 4 lines x 128 columns per workgroup when the pattern has a stride S of middle offsets
    // (+-n of a 3-D stencil) that 128 divides -- the four wavefronts' +-S gathers then hit lines their neighbours' diagonal gathers bring
    // into the same L1: 512^3 0.58 -> 0.49 ms --; otherwise contiguous 512-row chunks, each XCD walking runs of 8 of them (one L2 serves
    // the chunks' shared x lines: 0.58 -> 0.535 ms).  Variant bit28 forces the plain chunks round-robin (tests of that path at small sizes).
    // (Measured and dropped, profiles/EXPERIMENTS.md: tiles of 32 / 64 / 256 columns, XCD regions of the planes, four rows per lane, 512 / 1024 lanes per
    // workgroup, kernarg preload of the arguments.)
    const bool plain = (g_variant & 0x10000000) != 0;
    TL = DomTile{0, 0, 0, 0};
    long long wgs = (rows + 511) / 512;
    if (!plain) {
        const int S = dom_stride(P->dom);
        constexpr int cshift = 6;                         // 64 pairs per tile line: 128 columns, 4 lines
        constexpr int C = 2 << cshift, T = 256 >> cshift;
        if (S >= C && S % C == 0 && rows >= (long long)T * S) {
            const long long groups = rows / ((long long)T * S);
            TL.S = S; TL.cshift = cshift; TL.ntiled = (int)(groups * (S / C)); TL.nfull = (int)(groups * T * S);
            wgs = TL.ntiled + (rows - TL.nfull + 511) / 512;
        }
    }
    run = (plain || TL.S || wgs < 8 * NUM_XCD) ? 1 : 8;
    return wgs;
}

// which form the whole-matrix product of this plan takes: 0 the gathering kernels, 1 the z-marching kernel with the faces' masks, 2 its BOX form (x and y alone are streamed)
extern "C" int liship_csr_plan_marching(liship_csr_plan_t p)
{
    if (!p || !p->rowpat || !p->ptab8 || !p->vrec || !p->drec || p->products || !g_row_values || !g_row_patterns || !g_index_codes) return 0;
    LaunchArgs a{};
    a.plan = p; a.rb = 0; a.re = p->n; a.y = reinterpret_cast<double *>(16);
    DomMarch M;
    if (!dom_march_shape(a, M)) return 0;
    return ((M.order & 3) != 2 && g_dom_march != 3 && M.z0 >= p->box_z0 && M.z1 <= p->box_z1) ? 2 : 1;
}

// A launch over the rows [a.rb, a.re) in up to three parts: the planes the marching kernel serves (all of them, or the launch's share of the plan's box planes when that is
// most of it -- DIA's first and last plane differ from the others, a multi-rank slab's boundary planes are foreign) and what is left before and behind them, which the
// gathering kernel takes.  The parts' partial sums (fused dots) sit one after the other: a partial per workgroup, dom_shape() = their number.
struct DomPart { int rb, re; bool march; long long wgs; DomMarch M; DomTile TL; int run; };
static int dom_parts(const LaunchArgs &a, DomPart (&parts)[3])
{
    const liship_csr_plan_s *P = a.plan;
    auto gather = [&](int rb, int re) {
        DomPart G{rb, re, false, 0, DomMarch{}, DomTile{0, 0, 0, 0}, 1};
        LaunchArgs b = a; b.rb = rb; b.re = re;
        G.wgs = dom_gather_shape(b, G.TL, G.run);
        return G;
    };
    DomMarch M;
    if (dom_march_shape(a, M)) { parts[0] = DomPart{a.rb, a.re, true, M.wgs, M, DomTile{0, 0, 0, 0}, 1}; return 1; }
    int S = 0, SO = 0, perm = 0;
    if (g_dom_march && P && P->box_z1 > P->box_z0 && dom_seven_point(P->dom, S, SO, perm)) {
        const long long zlo = std::max<long long>(((long long)a.rb + SO - 1) / SO, P->box_z0), zhi = std::min<long long>((long long)a.re / SO, P->box_z1);
        if (zhi - zlo >= 8 && (zhi - zlo) * SO * 2 >= (long long)a.re - a.rb) {
            LaunchArgs b = a; b.rb = (int)(zlo * SO); b.re = (int)(zhi * SO);
            if (dom_march_shape(b, M)) {
                int np = 0;
                if (a.rb < b.rb) parts[np++] = gather(a.rb, b.rb);
                parts[np++] = DomPart{b.rb, b.re, true, M.wgs, M, DomTile{0, 0, 0, 0}, 1};
                if (b.re < a.re) parts[np++] = gather(b.re, a.re);
                return np;
            }
        }
    }
    parts[0] = gather(a.rb, a.re);
    return 1;
}
static long long dom_shape(const LaunchArgs &a, DomTile &TL, int &run)      // the number of workgroups with rows = partial sums of the fused forms (TL, run: the gathering kernel's, when it is the only part)
{
    DomPart parts[3];
    const int np = dom_parts(a, parts);
    long long total = 0;
    for (int i = 0; i < np; i++) total += parts[i].wgs;
    TL = parts[0].TL; run = parts[0].run;
    return total;
}

static void launch_dom(const LaunchArgs &a0, int dot = 0, const double *w = nullptr, double *partial0 = nullptr, const double *guard = nullptr, int pstride0 = 0)
{
    const liship_csr_plan_s *P = a0.plan;
    DomPart parts[3];
    const int np = dom_parts(a0, parts);
    long long total = 0;
    for (int i = 0; i < np; i++) total += parts[i].wgs;
    const int pstride = pstride0 ? pstride0 : (int)total;     // the second sums sit `stride` behind the first ones: the same for every part
    long long done = 0;
    for (int ip = 0; ip < np; ip++) {
        const DomPart &Q = parts[ip];
        LaunchArgs a = a0; a.rb = Q.rb; a.re = Q.re;
        double *partial = partial0 ? partial0 + done : nullptr;
        done += Q.wgs;
        if (Q.wgs <= 0) continue;
        if (Q.march) {                                // whole planes of a 7-point grid: the z-marching form
            const DomMarch &M = Q.M;
            const bool ws = dot != 0 && w != a.x;       // w is a vector of its own (else the diagonal's pair serves)
            const int ord = M.order & 3;
            const bool box = ord != 2 && g_dom_march != 3 && M.z0 >= P->box_z0 && M.z1 <= P->box_z1;      // (3: the masks' form on a box too, A/B)
            bool alt = false;
            for (int k = 0; k < 7; k++) alt = alt || ((P->box_modes >> (2 * k)) & 3) == 2;
            BoxAlt BA;
            for (int u = 0; u < 7; u++) BA.v[u] = P->box_alt[u];
            const bool pt = M.S % 128 != 0 || (M.SO / M.S) % 8 != 0;      // partial tiles (the last tile of a line / of a plane)
#define MARCH_ARGS a.rowpat, a.vrec, P->drec, P->dom, a.x, a.y, a.acc0, M, P->dom_xlen, w, partial, guard, pstride
#define GOM(DT, WS_, ORD_, GEN_) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, GEN_><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS)
#define GOMB(DT, WS_, ORD_) do { if (alt && pt) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true, true, false><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS, BA); \
                                 else if (alt) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true, true, false, false><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS, BA); \
                                 else if (P->box_pads && pt) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true, false, true><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS); \
                                 else if (P->box_pads) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true, false, true, false><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS); \
                                 else if (pt) spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS); \
                                 else spmv_csr_valuerec_march_kernel<2, 2, DT, WS_, ORD_, false, true, false, false, false><<<M.wgs, 256, 0, a.st>>>(MARCH_ARGS); } while (0)      /* whole tiles, the plain box form: no partial-tile tests */
#define GOMO(DT, WS_) do { if (box) { if (ord == 0) GOMB(DT, WS_, 0); else GOMB(DT, WS_, 1); } \
                           else if (!P->dom_simple || ord == 2) GOM(DT, WS_, 2, true); else if (ord == 0) GOM(DT, WS_, 0, false); else GOM(DT, WS_, 1, false); } while (0)
            if (dot == 0) GOMO(0, false); else if (dot == 1) { if (ws) GOMO(1, true); else GOMO(1, false); } else { if (ws) GOMO(2, true); else GOMO(2, false); }
#undef GOMO
#undef GOMB
#undef GOM
#undef MARCH_ARGS
            continue;
        }
        const DomTile TL = Q.TL;
        const int run = Q.run;
        const long long wgs = Q.wgs;
        const int span = NUM_XCD * run;
        const DomRec &DD = P->dom;
        const unsigned grid = (unsigned)((wgs + span - 1) / span * span);
        int wslot = -1;
        if (dot != 0 && w == a.x)
            for (int u = 0; u < 7; u++) if (((P->dom.mask >> u) & 1) && P->dom.off[u] == 0) wslot = u;
#define GOD(DT) spmv_csr_valuerec_dom_kernel<256, DT><<<grid, 256, 0, a.st>>>( \
            a.rowpat, a.vrec, P->drec, DD, P->dom_lo, P->dom_hi, a.x, a.y, Rows{a.rb, a.re, a.acc0}, run, TL, w, partial, guard, np > 1 ? pstride : pstride0, wslot, (int)wgs)
        if (dot == 0) GOD(0); else if (dot == 1) GOD(1); else GOD(2);
#undef GOD
    }
}

// the block-local kernel in the form the plan was built for: geometry 5 = round 3 (positions through LDS), 7 / 8 = positions in registers, x stage as long as the longest list
template <int G, int DOT>
void launch_local(const LaunchArgs &a, const double *w, double *partial, const double *guard, int pstride)
{
    constexpr Geometry g = kGeom[is_local_geom(G) ? G : LOCAL_GEOM];
    const int ndpl = a.plan ? a.plan->ndpl : 2, xcap = a.plan ? a.plan->xcap : 1024;
    const bool runs = a.plan && a.plan->drun && a.plan->droff && g_local_runs;      // lists of triples: the positions-in-registers forms read the run starts
#define GOL(NDPL_, XCAP_, RPOS_) do { if (RPOS_ && g_local_pairs) spmv_csr_local_kernel<g.block, g.work, DOT, NDPL_, XCAP_, RPOS_, false, RPOS_><<<a.nb, g.block, 0, a.st>>>( \
        a.ptr, a.idx, a.val, a.lcol, a.dcol, a.doff, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, w, partial, guard, pstride, g_uniform_rows, nullptr, nullptr, a.rowmap); \
    else spmv_csr_local_kernel<g.block, g.work, DOT, NDPL_, XCAP_, RPOS_><<<a.nb, g.block, 0, a.st>>>( \
        a.ptr, a.idx, a.val, a.lcol, a.dcol, a.doff, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, w, partial, guard, pstride, g_uniform_rows, nullptr, nullptr, a.rowmap); } while (0)
#define GOLR(NDPL_, XCAP_) do { if (g_local_pairs) spmv_csr_local_kernel<g.block, g.work, DOT, NDPL_, XCAP_, true, true, true><<<a.nb, g.block, 0, a.st>>>( \
        a.ptr, a.idx, a.val, a.lcol, a.dcol, a.doff, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, w, partial, guard, pstride, g_uniform_rows, a.plan->drun, a.plan->droff, a.rowmap); \
    else spmv_csr_local_kernel<g.block, g.work, DOT, NDPL_, XCAP_, true, true><<<a.nb, g.block, 0, a.st>>>( \
        a.ptr, a.idx, a.val, a.lcol, a.dcol, a.doff, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, w, partial, guard, pstride, g_uniform_rows, a.plan->drun, a.plan->droff, a.rowmap); } while (0)
    if constexpr (G == LOCAL_GEOM_R) { if (ndpl == 4) { if (runs) GOLR(4, 2048); else GOL(4, 2048, true); } else { if (runs) GOLR(2, 1024); else GOL(2, 1024, true); } }
    else if constexpr (G == LOCAL_GEOM4) {
        if (xcap <= 1024) { if (runs) GOLR(2, 1024); else GOL(2, 1024, true); }
        else if (xcap <= 1536) { if (runs) GOLR(4, 1536); else GOL(4, 1536, true); }
        else { if (runs) GOLR(4, 2048); else GOL(4, 2048, true); }
    } else { if (ndpl == 4) GOL(4, 2048, false); else GOL(2, 1024, false); }
#undef GOLR
    (void)xcap;
#undef GOL
}

template <int G>
void launch_geom(const LaunchArgs &a, int unroll, bool plan_products, int batch)
{
    const bool val16 = aligned16(a.val), idx16 = aligned16(a.idx);
    const bool idx8 = (reinterpret_cast<uintptr_t>(a.idx) & 7u) == 0;
    const bool products = plan_products || (g_variant & 6) != 0 || !(val16 && idx16);
    if (a.lcol && plan_products && is_local_geom(G) && g_variant == 0 && val16) {     // long rows, few distinct columns per row block
        launch_local<G, 0>(a, nullptr, nullptr, nullptr, 0);
        return;
    }
    if constexpr (G == LOCAL_GEOM4 || G == LOCAL_GEOM_R) {      // geometries only block-local plans have: whatever else such a plan launches (lists switched off,
        const bool vec = val16 && idx8;                          // values not 16 B aligned) takes the products kernel on the 4 B indices
        if (!vec) launch_products<G, 0>(a);
        else if (batch == 2) launch_products<G, 2>(a);
        else launch_products<G, 4>(a);
        return;
    } else {
    if (products) {
        const bool vec = !(g_variant & 2) && val16 && idx8;        // (variant bit 1: the scalar-load form on aligned arrays too -- tests)
        if (!vec) launch_products<G, 0>(a);
        else if (batch == 2) launch_products<G, 2>(a);
        else           launch_products<G, 4>(a);
        return;
    }
    const int U = unroll;
    if (a.rowpat && a.ptab8 && a.vrec && (g_variant & ~0x30004000) == 0) {            // the rows' values ride in the pattern records: one byte per row
        constexpr Geometry g = kGeom[G];
        const int chunks = (a.re - a.rb + g.block - 1) / g.block;       // rows [rb, re) in chunks of one workgroup's lanes, two per workgroup
        // beyond the Infinity Cache (256 MB of x) the two-rows-per-lane form wins (16 B requests: 320^3 +4 %, 448^3 +13 %, 512^3 +7 %);
        // below it x stays cache-resident from product to product and the one-row form is 15 % faster (tools/valuerec_probe.py)
        const bool pairs = (g_variant & 0x4000) || (long long)(a.re - a.rb) * 8 > (256ll << 20);
        // one pattern carries most rows: its gathers are issued together with the pattern bytes (one round trip, no LDS, no barrier)
        if (chunks > 0 && a.plan && a.plan->drec && !(g_variant & 0x20000000)) { launch_dom(a); return; }
        if (chunks > 0 && pairs)
            spmv_csr_valuerec_pair_kernel<g.block, 1><<<(chunks + 1) / 2, g.block, 0, a.st>>>(a.rowpat, a.vrec, a.npat1 - 1, a.x, a.y, Rows{a.rb, a.re, a.acc0});
        else if (chunks > 0)
            spmv_csr_valuerec_kernel<g.block, 2, 0><<<(chunks + 1) / 2, g.block, 0, a.st>>>(
                a.rowpat, a.vrec, a.npat1 - 1, a.x, a.y, a.blk, a.bfirst, chunks, Rows{a.rb, a.re, a.acc0});
        return;
    }
    if (a.rowpat && a.vrecw && (g_variant & ~0x4000) == 0 && launch_wide(a, nullptr)) return;      // wide records, x staged, the dominant pattern in scalar registers
    if (a.rowpat && a.vrecw && (g_variant & ~0x4000) == 0) {      // the rows' values ride in WIDE records (rows of up to 32 entries): one byte per row
        constexpr Geometry g = kGeom[G];
        spmv_csr_valuerecw_kernel<g.block, 0><<<a.nb, g.block, (size_t)(a.npat1 - 1) * (sizeof(double) * PATW_LEN + sizeof(int) * PATW_OFF), a.st>>>(
            a.rowpat, a.vrecw, a.npat1 - 1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0});
        return;
    }
    if (a.rowpat && a.ptab8 && (g_variant & ~0x20000000) == 0) {      // patterns of at most 7 offsets: gathers ahead of the slice
        // (round 3, measured and dropped: a wavefront per 64-row line segment, four lines per workgroup, the dominant pattern's gathers issued
        //  with the pattern bytes -- the shape that paid for the value records -- is bit-identical and 1.2 % faster at 512^3, 1.842 -> 1.821 ms:
        //  this kernel is bound by its value stream, not by x: profiles/r03_valuerec_dom_experiments.txt)
        constexpr Geometry g = kGeom[G];
        spmv_csr_pattern7_kernel<g.block, g.work, 0><<<a.nb, g.block, 0, a.st>>>(
            a.ptr, a.val, a.rowpat, a.rowrel, a.ptab8, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, nullptr, nullptr, nullptr, 0, xcd_strips(a));
        return;
    }
    if (a.rowpat && a.plan && a.plan->prec36 && g_team && (g_variant & ~0x4000) == 0) {    // patterns of 8..32 offsets, values streamed: four lanes per row (0x2000: the general kernel, A/B)
        // (measured and dropped, profiles/r03_pattern_team_kernel.txt: XCD slabs / runs of 1024+ workgroups +-2 %; the pattern byte speculated 2 %)
        launch_team(a, nullptr);
        return;
    }
    if (a.rowpat && a.ptab_len <= PAT_TABLE && (g_variant & ~0x2000) == 0) {    // one byte per ROW (the plan found <= 255 row patterns); 0x2000: experiment, table in LDS even for short patterns
        constexpr Geometry g = kGeom[G];
#define GOP(UU) spmv_csr_pattern_kernel<g.block, g.work, UU, 0><<<a.nb, g.block, 0, a.st>>>( \
            a.ptr, a.idx, a.val, a.rowpat, a.rowrel, a.ptab, a.ptab_len, a.npat1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz)
        if (U == 4) GOP(4); else if (U == 7) GOP(7); else GOP(8);
#undef GOP
        return;
    }
    if (a.codes && (g_variant & ~0xF0) == 0) {     // one-byte column codes (the plan found <= 255 diagonals)
        constexpr Geometry g = kGeom[G];
#define GO(UU) spmv_csr_coded_kernel<g.block, g.work, UU, 0><<<a.nb, g.block, 0, a.st>>>( \
            a.ptr, a.idx, a.val, a.codes, a.dict, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, nullptr, nullptr, nullptr, 0, xcd_strips(a))
        if (U == 4) GO(4); else if (U == 7) GO(7); else GO(8);
#undef GO
        return;
    }
    if (U == 4)      launch_rowgather<G, 4>(a);
    else if (U == 7) launch_rowgather<G, 7>(a);
    else             launch_rowgather<G, 8>(a);
    }
}

template <int G, int DOT>
void launch_rowgather_dot(const LaunchArgs &a, int unroll, const double *w, double *partial, int pstride = 0)
{
    constexpr Geometry g = kGeom[G];
    if (a.rowpat && a.ptab8 && a.vrec && a.plan && a.plan->drec && !(g_variant & 0x20002000) && g.block == 256) {    // the dominant pattern speculated (see launch_geom)
        const liship_csr_plan_s *P = a.plan;
        {                                                 // four rows per lane, a wavefront per row block
            int wslot = -1;
            if (w == a.x)
                for (int u = 0; u < 7; u++) if (((P->dom.mask >> u) & 1) && P->dom.off[u] == 0) wslot = u;
            spmv_csr_valuerec_dom_dot4_kernel<256, DOT><<<(a.nb + 3) / 4, 256, 0, a.st>>>(
                a.rowpat, a.vrec, P->drec, P->dom, P->dom_lo, P->dom_hi, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0},
                w, partial, liship_internal_guard(), pstride, wslot);
            return;
        }
    }
    if (a.rowpat && a.ptab8 && a.vrec && !(g_variant & 0x2000) && g.block == 256 &&
        ((g_variant & 0x4000) || (long long)(a.re - a.rb) * 8 > (256ll << 20))) {          // two rows per lane beyond the Infinity Cache (see launch_geom)
        spmv_csr_valuerec_pair_dot_kernel<256, DOT><<<(a.nb + 1) / 2, 256, 0, a.st>>>(
            a.rowpat, a.vrec, a.npat1 - 1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0},
            w, partial, liship_internal_guard(), pstride);
        return;
    }
    if (a.rowpat && a.ptab8 && a.vrec && !(g_variant & 0x2000)) {
        spmv_csr_valuerec_kernel<g.block, 2, DOT><<<(a.nb + 1) / 2, g.block, 0, a.st>>>(
            a.rowpat, a.vrec, a.npat1 - 1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0},
            w, partial, liship_internal_guard(), pstride);
        return;
    }
    if (a.rowpat && a.ptab8 && !(g_variant & 0x2000)) {
        spmv_csr_pattern7_kernel<g.block, g.work, DOT><<<a.nb, g.block, 0, a.st>>>(
            a.ptr, a.val, a.rowpat, a.rowrel, a.ptab8, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz,
            w, partial, liship_internal_guard(), pstride, xcd_strips(a));
        return;
    }
    if (a.rowpat && a.vrecw && !(g_variant & 0x2000)) {
        spmv_csr_valuerecw_kernel<g.block, DOT><<<a.nb, g.block, (size_t)(a.npat1 - 1) * (sizeof(double) * PATW_LEN + sizeof(int) * PATW_OFF), a.st>>>(
            a.rowpat, a.vrecw, a.npat1 - 1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0},
            w, partial, liship_internal_guard(), pstride);
        return;
    }
    if (a.rowpat && a.ptab_len <= PAT_TABLE) {
#define GOP(UU) spmv_csr_pattern_kernel<g.block, g.work, UU, DOT><<<a.nb, g.block, 0, a.st>>>( \
            a.ptr, a.idx, a.val, a.rowpat, a.rowrel, a.ptab, a.ptab_len, a.npat1, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, \
            w, partial, liship_internal_guard(), pstride)
        if (unroll == 4) GOP(4); else if (unroll == 7) GOP(7); else GOP(8);
#undef GOP
        return;
    }
    if (a.codes) {
#define GO(UU) spmv_csr_coded_kernel<g.block, g.work, UU, DOT><<<a.nb, g.block, 0, a.st>>>( \
            a.ptr, a.idx, a.val, a.codes, a.dict, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, w, partial, liship_internal_guard(), pstride, xcd_strips(a))
        if (unroll == 4) GO(4); else if (unroll == 7) GO(7); else GO(8);
#undef GO
        return;
    }
#define GO(UU) spmv_csr_rowgather_kernel<g.block, g.work, UU, DOT><<<a.nb, g.block, 0, a.st>>>( \
        a.ptr, a.idx, a.val, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, a.nnz, xcd_strips(a), w, partial, liship_internal_guard(), pstride)
    if (unroll == 4) GO(4); else if (unroll == 7) GO(7); else GO(8);
#undef GO
}

template <int G, int DOT>
void launch_products_dot(const LaunchArgs &a, int batch, const double *w, double *partial, int pstride = 0)
{
    constexpr Geometry g = kGeom[G];
    if (a.lcol && is_local_geom(G)) {
        launch_local<G, DOT>(a, w, partial, liship_internal_guard(), pstride);
        return;
    }
    if (batch == 2)
        spmv_csr_products_kernel<g.block, g.work, 2, DOT>
            <<<a.nb, g.block, 0, a.st>>>(a.ptr, a.idx, a.val, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, w, partial, liship_internal_guard(), pstride, a.order);
    else
        spmv_csr_products_kernel<g.block, g.work, 4, DOT>
            <<<a.nb, g.block, 0, a.st>>>(a.ptr, a.idx, a.val, a.x, a.y, a.blk, a.bfirst, a.nb, Rows{a.rb, a.re, a.acc0}, w, partial, liship_internal_guard(), pstride, a.order);
}

// tree mode (liship_spmv_csr_set_long_row_tree), plans with row blocks of more than TAIL_FROM entries: the tails of those blocks' last rows are summed by a
// workgroup per TAIL_CHUNK entries and left in y[row] for the product's kernel (block_by_products) -- in front of EVERY launch of such a plan, on its row range
static void tail_prepass(const liship_csr_plan_s *p, const LaunchArgs &a)
{
    if (!g_long_row_tree_host || !p || p->ntchunk <= 0 || a.nb <= 0) return;
    const Rows RW{a.rb, a.re, a.acc0};
    spmv_csr_tail_chunks_kernel<256><<<p->ntchunk, 256, 0, a.st>>>(a.ptr, a.idx, a.val, a.x, a.blk, p->tchunk, RW, p->tpart);
    spmv_csr_tail_fold_kernel<<<(p->nheavy + 63) / 64, 64, 0, a.st>>>(p->nheavy, a.ptr, a.blk, p->thead, RW, p->tpart, a.y, a.rowmap);
}

int launch_csr(liship_csr_plan_t p, const LaunchArgs &a0)
{
    if (a0.nb <= 0) return 0;
    LaunchArgs a = a0;
    a.plan = p;
    tail_prepass(p, a);
    switch (p->geom) {
        case 0: launch_geom<0>(a, p->unroll, p->products != 0, p->batch); break;
        case 1: launch_geom<1>(a, p->unroll, p->products != 0, p->batch); break;
        case 5: launch_geom<5>(a, p->unroll, p->products != 0, p->batch); break;
        case 7: launch_geom<7>(a, p->unroll, p->products != 0, p->batch); break;
        case 8: launch_geom<8>(a, p->unroll, p->products != 0, p->batch); break;
        default: return LISHIP_ERR_ARG;
    }
    LAUNCH_CHECK();
    return 0;
}

} // namespace

// Does the product of this plan run one of the kernels with a row split of their own (16 / 64 rows per wavefront: the team and staged kernels) under the
// switches in force?  Their launches have no per-row-block epilogue: the fused entry points below refuse (LISHIP_ERR_ARG) and the caller runs the product
// and one reduction pass -- measured FASTER than the product followed by a pass that rebuilt the row blocks' partial sums (BiCGSTAB at 160^3 2290 against
// 2125 it/s, profiles/r03_pattern_team_kernel.txt).
static bool plan_runs_wide(const liship_csr_plan_s *p, int rb, int re)      // the staged wide-record kernels: an epilogue of their own (one partial per workgroup of 256 rows / block rows)
{
    return p && p->rowpat && g_row_patterns && g_index_codes && g_team && !p->ptab8 && g_row_values && p->vrecw && g_variant == 0 &&
           ((p->wdrec && p->wstage && p->wd.len > 0) || block_rows_serve(p, rb, re));
}
static bool plan_runs_teams(const liship_csr_plan_s *p)     // the four-lanes-per-row kernels (values streamed): no epilogue
{
    if (!p || !p->rowpat || !g_row_patterns || !g_index_codes || !g_team || p->ptab8) return false;
    if (g_row_values && p->vrecw) return false;
    return p->prec36 != nullptr && (g_variant & ~0x4000) == 0;
}
static bool plan_runs_dom(const liship_csr_plan_s *p)       // the dominant-pattern product of a plan with value records: its tiles have an epilogue of their own too
{                                                           // (variant 0x4000: the fused dots stay with the row blocks' partial sums, spmv_csr_valuerec_dom_dot4_kernel -- A/B, tests)
    return p && p->rowpat && g_row_patterns && g_index_codes && p->ptab8 && g_row_values && p->vrec && p->drec && !p->products &&
           kGeom[p->geom].block == 256 && (g_variant & ~0x10000000) == 0 && !g_row_block_dots;
}
// the reordered form serves whole-matrix products of the plan in its shipped configuration; row ranges and the fused reductions (whose partial sums follow the
// ORIGINAL row blocks) keep the original numbering
static bool plan_has_reordered_form(const liship_csr_plan_s *p)      // ... for callers that iterate in the new numbering (liship_csr_plan_reordered_form)
{
    if (!p || !p->inner || !g_reorder || g_variant != 0) return false;
    return p->inner->products ? (p->inner->lcol && g_local_cols) : true;
}
// ... and for single products: long rows only.  A product in the caller's numbering pays a gather of x and a scattered store of y, one random access per node each:
// with 3 unknowns per node and ~70 entries per row that is 8 % of the product (Queen class), with scalar unknowns and 7 entries per row it is four times the product
// (tools/scrambled_short_rows_probe.py: 160^3 7-point, numbered at random inside runs of 4096: 0.130 ms as it is, 0.257 ms renumbered per product -- and 0.06 inside a solve)
// Opt-in even there (liship_spmv_csr_set_reorder(2), LIS_AMD_REORDER_PRODUCTS=1): over six boxes the Queen-class product moved between -5.5 % and +1.5 % -- the
// two passes eat what the kernel gains (0.636 -> 0.58-0.60 ms), and the spread of a box's page placement is as large as the rest.  The loops take the whole gain.
static bool plan_runs_reordered(const liship_csr_plan_s *p) { return g_reorder == 2 && plan_has_reordered_form(p) && p->products && p->inner->products && p->ncols <= p->n; }      // (a rank's local matrix: x carries ghost entries behind the rows -- its single products keep the caller's numbering)
static int launch_reordered(liship_csr_plan_t p, const double *x, double *y, hipStream_t st)
{
    const liship_csr_plan_s *q = p->inner;
    csr_reorder_gather_kernel<<<(p->n / 3 + 256) / 256, 256, 0, st>>>(p->n, p->r_perm, x, p->r_x);
    LaunchArgs a{p->r_ptr, p->r_idx, p->r_val, p->r_x, y, q->blk, 0, q->nblocks, 0, q->n, (int)q->nnz, st, nullptr, nullptr, q->lcol, q->dcol, q->doff, p->first_term ? -0.0 : 0.0};
    a.rowmap = p->r_perm;
    return launch_csr(p->inner, a);
}
extern "C" int liship_csr_plan_fused_dots(liship_csr_plan_t p) { return (plan_runs_teams(p) || plan_runs_reordered(p)) ? 0 : 1; }
// The reordered form as a matrix of its own -- P A P^T: its plan (owned by `p`), its arrays, the permutation (new position -> original row) -- for a caller that keeps
// whole iterations in the new numbering (lis_solve: b and x0 gathered once, every product, dot and update on renumbered vectors, x scattered back at the end).
extern "C" int liship_csr_plan_reordered_form(liship_csr_plan_t p, liship_csr_plan_t *inner, const int **ptr, const int **idx, const double **val, const int **perm)
{
    if (!plan_has_reordered_form(p) || !inner || !ptr || !idx || !val || !perm) return LISHIP_ERR_ARG;
    *inner = p->inner; *ptr = p->r_ptr; *idx = p->r_idx; *val = p->r_val; *perm = p->r_perm;
    return 0;
}
extern "C" int liship_permute_gather_f64(int n, const int *perm, const double *x, double *xp, void *stream)       // xp[i] = x[perm[i]]
{
    if (n < 0 || (n > 0 && (!perm || !x || !xp || x == xp))) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    csr_reorder_gather_kernel<<<(n / 3 + 256) / 256, 256, 0, as_stream(stream)>>>(n, perm, x, xp);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int liship_permute_scatter_f64(int n, const int *perm, const double *xp, double *x, void *stream)      // x[perm[i]] = xp[i]
{
    if (n < 0 || (n > 0 && (!perm || !x || !xp || x == xp))) return LISHIP_ERR_ARG;
    if (n == 0) return 0;
    csr_reorder_scatter_kernel<<<(n / 3 + 256) / 256, 256, 0, as_stream(stream)>>>(n, perm, xp, x);
    LAUNCH_CHECK();
    return 0;
}
namespace {
__global__ void rows_of_list_kernel(int count, const int *__restrict__ inv, const int *__restrict__ index, int *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = inv[index[k]];
}
}
extern "C" int liship_permute_rows_of_list(int n, const int *perm, int count, const int *index, int *out, void *stream)      // out[k] = new position of row index[k]
{
    if (n < 0 || count < 0 || (n > 0 && !perm) || (count > 0 && (!index || !out))) return LISHIP_ERR_ARG;
    if (count == 0 || n == 0) return 0;
    hipStream_t st = as_stream(stream);
    int *inv = nullptr;
    HIP_TRY(hipMalloc(&inv, sizeof(int) * (size_t)n));
    csr_reorder_inverse<<<(n + 255) / 256, 256, 0, st>>>(n, perm, inv);
    rows_of_list_kernel<<<(count + 255) / 256, 256, 0, st>>>(count, inv, index, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(inv);
    return e == hipSuccess ? 0 : (int)e;
}
// upper bound of the partial-sum slots the fused product needs when it is launched in up to three row ranges (liship_spmv_csr_rows_dot_f64)
extern "C" long long liship_csr_plan_fused_slots(liship_csr_plan_t p)
{
    if (!p) return 0;
    if (plan_runs_dom(p)) return ((long long)p->n + 511) / 512 + 3 * 16;                // a partial per workgroup (tile); every range ends in less than one group of tiles
    if (plan_runs_wide(p, 0, p->n)) {                   // (block rows only: a range that cuts a block row runs the row blocks' kernel)
        const long long wide = ((long long)p->n + 255) / 256 + 3, blocks = (long long)p->nblocks + 2;
        return (p->wdrec && p->wd.len > 0) || wide > blocks ? wide : blocks;
    }
    return (long long)p->nblocks + 2;
}

extern "C" int liship_csr_plan_block2_march(liship_csr_plan_t p)
{
    if (!p || p->b2.S <= 0 || !block_rows_serve(p, 0, p->n)) return 0;
    LaunchArgs a{};
    a.plan = p; a.rb = 0; a.re = p->n; a.y = reinterpret_cast<double *>(16); a.acc0 = p->first_term ? -0.0 : 0.0;
    Block2March M;
    return block2_shape(a, M) ? 1 : 0;
}
extern "C" int liship_csr_plan_box27(liship_csr_plan_t p)
{
    if (!p || p->b27.S <= 0 || !p->vrecw || !g_team || !g_row_patterns || !g_index_codes) return 0;
    LaunchArgs a{};
    a.plan = p; a.rb = 0; a.re = p->n; a.y = reinterpret_cast<double *>(16); a.acc0 = p->first_term ? -0.0 : 0.0;
    Box27 M;
    int lpw = 0;
    return box27_shape(a, M, lpw) ? 1 : 0;
}

extern "C" int liship_spmv_csr_f64(liship_csr_plan_t p, const int *ptr, const int *idx,
                                   const double *val, const double *x, double *y, void *stream)
{
    if (!p) return LISHIP_ERR_ARG;
    if (plan_runs_reordered(p) && x != y) return launch_reordered(p, x, y, as_stream(stream));
    LaunchArgs a{ptr, idx, val, x, y, p->blk, 0, p->nblocks, 0, p->n, (int)p->nnz, as_stream(stream), g_index_codes ? p->codes : nullptr, p->dict, g_local_cols ? p->lcol : nullptr, p->dcol, p->doff, p->first_term ? -0.0 : 0.0, (g_row_patterns && g_index_codes) ? p->rowpat : nullptr, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, p->ptab8, g_row_values ? p->vrec : nullptr, p->order, g_row_values ? p->vrecw : nullptr};
    return launch_csr(p, a);
}

// y = A x and, in the same pass, result[0] = sum_r w[r]*y[r] (w may be x), result[1] = sum_r y[r]^2 if want_sumsq.
// `work` is the reduction scratch (liship_reduce_work_bytes).  LISHIP_ERR_ARG when the fused form cannot
// serve the call (unaligned arrays, more row blocks than scratch slots, non-default geometry): use the plain
// product + liship_dot_f64 then.
extern "C" int liship_spmv_csr_dot_f64(liship_csr_plan_t p, const int *ptr, const int *idx, const double *val,
                                       const double *x, double *y, const double *w, int want_sumsq,
                                       double *result, void *work, void *stream)
{
    if (!p || !w || !result || !work) return LISHIP_ERR_ARG;
    if (plan_runs_teams(p) || plan_runs_reordered(p) || liship_internal_ref_chunks()) return LISHIP_ERR_ARG;      // (reference-order sums: the product, then one ordered pass)
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    if ((g_variant & ~0x30006000) != 0 || (size_t)p->nblocks > slots || !aligned16(val) || !aligned16(idx)) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    if (plan_runs_wide(p, 0, p->n) && p->n > 0 && (size_t)((p->n + 255) / 256) <= slots) {       // wide records, x staged: a partial per workgroup of 256 rows
        LaunchArgs aw{ptr, idx, val, x, y, p->blk, 0, p->nblocks, 0, p->n, (int)p->nnz, as_stream(stream), p->codes, p->dict, nullptr, nullptr, nullptr, p->first_term ? -0.0 : 0.0, p->rowpat, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, nullptr, nullptr, nullptr, p->vrecw};
        aw.plan = p;
        int wgs = 0;
        launch_wide(aw, liship_internal_guard(), want_sumsq ? 2 : 1, w, partial, 0, &wgs);
        LAUNCH_CHECK();
        return liship_internal_fold(wgs, want_sumsq ? 2 : 1, wgs, partial, spare, result, stream);
    }
    LaunchArgs a{ptr, idx, val, x, y, p->blk, 0, p->nblocks, 0, p->n, (int)p->nnz, as_stream(stream), g_index_codes ? p->codes : nullptr, p->dict, g_local_cols ? p->lcol : nullptr, p->dcol, p->doff, p->first_term ? -0.0 : 0.0, (g_row_patterns && g_index_codes) ? p->rowpat : nullptr, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, p->ptab8, g_row_values ? p->vrec : nullptr, p->order, g_row_values ? p->vrecw : nullptr};
    a.plan = p;
    tail_prepass(p, a);
    if (p->nblocks == 0) { HIP_TRY(hipMemsetAsync(result, 0, sizeof(double) * 2, a.st)); return 0; }
    if (plan_runs_dom(p)) {                            // value records, dominant pattern: a partial per workgroup (tile) of the plain product's shape
        DomTile TL;
        int run = 1;
        const long long wgs = dom_shape(a, TL, run);
        const long long np = wgs;                         // a partial per workgroup
        if (wgs > 0 && (size_t)np <= slots) {
            launch_dom(a, want_sumsq ? 2 : 1, w, partial, liship_internal_guard(), 0);
            LAUNCH_CHECK();
            return liship_internal_fold((int)np, want_sumsq ? 2 : 1, (int)np, partial, spare, result, stream);
        }
    }
    if (p->products && p->geom == LOCAL_GEOM) {      // a plan with block-local columns (or one that has them switched off)
        if (want_sumsq) launch_products_dot<LOCAL_GEOM, 2>(a, p->batch, w, partial); else launch_products_dot<LOCAL_GEOM, 1>(a, p->batch, w, partial);
    } else if (p->products && p->geom == LOCAL_GEOM4) {
        if (want_sumsq) launch_products_dot<LOCAL_GEOM4, 2>(a, p->batch, w, partial); else launch_products_dot<LOCAL_GEOM4, 1>(a, p->batch, w, partial);
    } else if (p->products && p->geom == LOCAL_GEOM_R) {
        if (want_sumsq) launch_products_dot<LOCAL_GEOM_R, 2>(a, p->batch, w, partial); else launch_products_dot<LOCAL_GEOM_R, 1>(a, p->batch, w, partial);
    } else if (p->products) {       // geometry 1
        if (want_sumsq) launch_products_dot<1, 2>(a, p->batch, w, partial); else launch_products_dot<1, 1>(a, p->batch, w, partial);
    } else if (p->geom == 1) { if (want_sumsq) launch_rowgather_dot<1, 2>(a, p->unroll, w, partial); else launch_rowgather_dot<1, 1>(a, p->unroll, w, partial); }
    else if (want_sumsq) launch_rowgather_dot<0, 2>(a, p->unroll, w, partial);
    else                 launch_rowgather_dot<0, 1>(a, p->unroll, w, partial);
    LAUNCH_CHECK();
    return liship_internal_fold(p->nblocks, want_sumsq ? 2 : 1, p->nblocks, partial, spare, result, stream);
}

extern "C" int liship_spmv_csr_rows_f64(liship_csr_plan_t p, int row_begin, int row_end, const int *ptr,
                                        const int *idx, const double *val, const double *x, double *y,
                                        void *stream)
{
    if (!p || row_begin < 0 || row_end > p->n) return LISHIP_ERR_ARG;
    if (row_begin >= row_end || p->nblocks == 0) return 0;
    // row blocks whose interval [blk[b].row, blk[b+1].row) intersects [row_begin,row_end)
    const v2i32 *br = p->blk_host;
    int lo = 0, hi = p->nblocks;                 // first b with br[b+1].row > row_begin
    while (lo < hi) { int mid = (lo + hi) / 2; if (br[mid + 1].x > row_begin) hi = mid; else lo = mid + 1; }
    const int bfirst = lo;
    lo = bfirst; hi = p->nblocks;                // first b with br[b].row >= row_end
    while (lo < hi) { int mid = (lo + hi) / 2; if (br[mid].x >= row_end) hi = mid; else lo = mid + 1; }
    LaunchArgs a{ptr, idx, val, x, y, p->blk, bfirst, lo - bfirst, row_begin, row_end, (int)p->nnz, as_stream(stream), g_index_codes ? p->codes : nullptr, p->dict, g_local_cols ? p->lcol : nullptr, p->dcol, p->doff, p->first_term ? -0.0 : 0.0, (g_row_patterns && g_index_codes) ? p->rowpat : nullptr, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, p->ptab8, g_row_values ? p->vrec : nullptr, nullptr, g_row_values ? p->vrecw : nullptr};
    return launch_csr(p, a);
}

// The fused-reduction product in parts (a multi-rank job runs the rows that reference no ghost column while the
// halo is in flight, the boundary rows after it): each part leaves one partial per row block it launched at
// work[slot_base ...) (second result `slots` further), liship_spmv_csr_dot_finish_f64 folds all of them.  The sum
// order is a function of the parts' row ranges only.  LISHIP_ERR_ARG as for liship_spmv_csr_dot_f64, or when the
// slots run out.
extern "C" int liship_spmv_csr_rows_dot_f64(liship_csr_plan_t p, int row_begin, int row_end, const int *ptr, const int *idx,
                                            const double *val, const double *x, double *y, const double *w, int want_sumsq,
                                            void *work, int slot_base, int *slots_used, void *stream)
{
    if (!p || !w || !work || !slots_used || row_begin < 0 || row_end > p->n || slot_base < 0) return LISHIP_ERR_ARG;
    if (plan_runs_teams(p) || liship_internal_ref_chunks()) return LISHIP_ERR_ARG;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    if ((g_variant & ~0x30006000) != 0 || !aligned16(val) || !aligned16(idx)) return LISHIP_ERR_ARG;
    *slots_used = 0;
    if (row_begin >= row_end || p->nblocks == 0) return 0;
    if (plan_runs_wide(p, row_begin, row_end)) {       // wide records, x staged: a partial per workgroup of 256 rows (block rows) of the range
        int wgs = (row_end - row_begin + 255) / 256;
        if ((size_t)slot_base + (size_t)wgs > slots) return LISHIP_ERR_ARG;
        LaunchArgs aw{ptr, idx, val, x, y, p->blk, 0, p->nblocks, row_begin, row_end, (int)p->nnz, as_stream(stream), p->codes, p->dict, nullptr, nullptr, nullptr, p->first_term ? -0.0 : 0.0, p->rowpat, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, nullptr, nullptr, nullptr, p->vrecw};
        aw.plan = p;
        launch_wide(aw, liship_internal_guard(), want_sumsq ? 2 : 1, w, static_cast<double *>(work) + slot_base, (int)slots, &wgs);
        LAUNCH_CHECK();
        *slots_used = wgs;
        return 0;
    }
    if (plan_runs_dom(p)) {                            // value records, dominant pattern: a partial per workgroup of the range's own tiles
        LaunchArgs ad{ptr, idx, val, x, y, p->blk, 0, p->nblocks, row_begin, row_end, (int)p->nnz, as_stream(stream), p->codes, p->dict, nullptr, nullptr, nullptr, p->first_term ? -0.0 : 0.0, p->rowpat, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, p->ptab8, p->vrec, nullptr, nullptr};
        ad.plan = p;
        DomTile TL;
        int run = 1;
        const long long wgs = dom_shape(ad, TL, run);
        const long long np = wgs;
        if ((size_t)slot_base + (size_t)np > slots) return LISHIP_ERR_ARG;
        launch_dom(ad, want_sumsq ? 2 : 1, w, static_cast<double *>(work) + slot_base, liship_internal_guard(), (int)slots);
        LAUNCH_CHECK();
        *slots_used = (int)np;
        return 0;
    }
    const v2i32 *br = p->blk_host;
    int lo = 0, hi = p->nblocks;                 // first b with br[b+1].row > row_begin
    while (lo < hi) { int mid = (lo + hi) / 2; if (br[mid + 1].x > row_begin) hi = mid; else lo = mid + 1; }
    const int bfirst = lo;
    lo = bfirst; hi = p->nblocks;                // first b with br[b].row >= row_end
    while (lo < hi) { int mid = (lo + hi) / 2; if (br[mid].x >= row_end) hi = mid; else lo = mid + 1; }
    const int nb = lo - bfirst;
    if (nb <= 0) return 0;
    if ((size_t)slot_base + (size_t)nb > slots) return LISHIP_ERR_ARG;
    double *partial = static_cast<double *>(work) + slot_base;
    LaunchArgs a{ptr, idx, val, x, y, p->blk, bfirst, nb, row_begin, row_end, (int)p->nnz, as_stream(stream), g_index_codes ? p->codes : nullptr, p->dict, g_local_cols ? p->lcol : nullptr, p->dcol, p->doff, p->first_term ? -0.0 : 0.0, (g_row_patterns && g_index_codes) ? p->rowpat : nullptr, p->rowrel, p->ptab, p->ptab_len, p->npat + 1, p->ptab8, g_row_values ? p->vrec : nullptr, nullptr, g_row_values ? p->vrecw : nullptr};
    a.plan = p;
    tail_prepass(p, a);
    const int ps = (int)slots;
    if (p->products && p->geom == LOCAL_GEOM) {
        if (want_sumsq) launch_products_dot<LOCAL_GEOM, 2>(a, p->batch, w, partial, ps); else launch_products_dot<LOCAL_GEOM, 1>(a, p->batch, w, partial, ps);
    } else if (p->products && p->geom == LOCAL_GEOM4) {
        if (want_sumsq) launch_products_dot<LOCAL_GEOM4, 2>(a, p->batch, w, partial, ps); else launch_products_dot<LOCAL_GEOM4, 1>(a, p->batch, w, partial, ps);
    } else if (p->products && p->geom == LOCAL_GEOM_R) {
        if (want_sumsq) launch_products_dot<LOCAL_GEOM_R, 2>(a, p->batch, w, partial, ps); else launch_products_dot<LOCAL_GEOM_R, 1>(a, p->batch, w, partial, ps);
    } else if (p->products) {
        if (want_sumsq) launch_products_dot<1, 2>(a, p->batch, w, partial, ps); else launch_products_dot<1, 1>(a, p->batch, w, partial, ps);
    } else if (p->geom == 1) { if (want_sumsq) launch_rowgather_dot<1, 2>(a, p->unroll, w, partial, ps); else launch_rowgather_dot<1, 1>(a, p->unroll, w, partial, ps); }
    else if (want_sumsq) launch_rowgather_dot<0, 2>(a, p->unroll, w, partial, ps);
    else                 launch_rowgather_dot<0, 1>(a, p->unroll, w, partial, ps);
    LAUNCH_CHECK();
    *slots_used = nb;
    return 0;
}

extern "C" int liship_spmv_csr_dot_finish_f64(int slots_used, int want_sumsq, double *result, void *work, void *stream)
{
    if (!result || !work || slots_used < 0) return LISHIP_ERR_ARG;
    const size_t slots = liship_reduce_work_bytes() / sizeof(double) / 4;
    if ((size_t)slots_used > slots) return LISHIP_ERR_ARG;
    if (slots_used == 0) { HIP_TRY(hipMemsetAsync(result, 0, sizeof(double) * 2, as_stream(stream))); return 0; }
    double *partial = static_cast<double *>(work), *spare = partial + 2 * slots;
    return liship_internal_fold(slots_used, want_sumsq ? 2 : 1, (int)slots, partial, spare, result, stream);
}
