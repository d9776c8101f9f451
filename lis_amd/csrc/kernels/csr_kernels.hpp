// csr_kernels.hpp -- the CSR product kernels of spmv_csr.hip (included there, once: one translation unit).  Geometry, the switches, the row-block helpers, then one
// kernel per plan form in the order they were built: products / row gather (the contract form) / block-local columns / coded / row patterns / team / pattern
// records / value records / dominant pattern / z-marching (7-point, 27-point, 2 x 2 blocks) / wide records / block rows.  Which plan runs which: DESIGN.md 4.
#pragma once
namespace {

// BLOCK threads per workgroup, WORK merge-path items (rows + non-zeros) per row block.
struct Geometry { int block, work; };
constexpr Geometry kGeom[] = {
    {192, 1408},   // 0: default -- 176 stencil rows per block, 18.5 KB LDS, 8 workgroups per CU
    {256, 2048},   // 1
    {128,  896},   // 2  (ids 2, 3, 4, 6: geometries of the round-1 sweep, no longer instantiated -- liship_csr_plan_create refuses them; the ids of the others stay)
    {192, 1536},   // 3
    {256, 1536},   // 4
    {512, 4096},   // 5
    { 64,  512},   // 6
    {512, 3072},   // 7: block-local columns, lists of up to 2048 columns, positions in registers: 39.5 KB of LDS (lists <= 1536), FOUR workgroups per CU
    {512, 3584},   // 8: block-local columns, lists of up to 1024 columns, positions in registers: 39.5 KB of LDS, FOUR workgroups per CU
};
constexpr int kNumGeom = sizeof(kGeom) / sizeof(kGeom[0]);
constexpr int LOCAL_GEOM = 5;    // block-local columns (spmv_csr_local_kernel), rounds 2-3: 512 lanes / 4096 items, positions staged in LDS (53 / 61 KB: 3 / 2 workgroups per CU)
// Round 4.  The kernel is bound by the loads a CU keeps in flight, and those by what its LDS can stage: rocprof's occupancy line showed the
// four-columns-per-lane form (Queen class: lists of ~1500 columns, 61 KB) running TWO workgroups per CU.  The 2 B positions never needed LDS -- a lane
// reads the positions of its own items only -- so they go straight to registers, and with the stage sized to 39.5 KB FOUR workgroups fit:
//   lists <= 1024 columns: 3584 items + 8 KB of x   (150 KB of loads in flight per CU instead of 129)
//   longer lists:          3072 items + 12 / 16 KB  (141 / 106 KB instead of 92)
constexpr int LOCAL_GEOM4 = 7, LOCAL_GEOM_R = 8;
constexpr bool is_local_geom(int g) { return g == LOCAL_GEOM || g == LOCAL_GEOM4 || g == LOCAL_GEOM_R; }
constexpr int ROW_ALIGN = 16;    // rows per 128 B line of y / x
constexpr int SLACK = 128;       // extra items a block may take to start on an aligned row

// Form selector (liship_spmv_csr_set_variant); 0 is the shipped configuration.  Every value gives the reference's bits: the bits choose among kernels
// that all compute the same sums in the same order (tests and A/B measurements select them by hand).
//   bit1  products kernel with scalar loads     bit2  products kernel with vector loads
//   bits4-7 geometry id and bit24 "no row alignment": read at plan creation
//   bit13 (0x2000) row patterns through the general kernel (table in LDS) even when the plan has 32 B records
//   bit14 (0x4000) value records: the two-rows-per-lane kernel whatever the size (tests; by default only beyond 256 MB of x)
//   bit29 (0x20000000) value records: never the dominant-pattern kernels (the kernels of plans without a dominant pattern: tests);
//   bit28 (0x10000000) the dominant-pattern product on contiguous chunks instead of tiles (the path of grids whose lines 128 does not divide: tests)
int g_variant = 0;
int g_index_codes = 1;           // liship_spmv_csr_set_index_codes: 0 keeps every product on the 4 B indices
int g_row_patterns = 1;          // liship_spmv_csr_set_row_patterns: 0 keeps coded matrices on one byte per non-zero
int g_row_values = 1;            // liship_spmv_csr_set_row_values: 0 keeps streaming the values of matrices that have value records
int g_local_cols = 1;            // liship_spmv_csr_set_local_columns: 0 keeps the products kernel on the 4 B indices
int g_xcd_strips = 1;            // liship_spmv_csr_set_xcd_strips: 0 keeps the row blocks of the 7-offset pattern kernel in their natural (round-robin over the XCDs) order
int g_local_rpos = 1;            // liship_spmv_csr_set_local_register_positions: 0 = plans built from now on take the round-3 form (4096-item blocks, positions through LDS)
int g_uniform_rows = 1;          // liship_spmv_csr_set_uniform_rows: 0 keeps the row sums of the block-local kernel on the skewed schedule everywhere (A/B)
int g_long_row_tree_host = 1;    // host mirror of d_long_row_tree (liship_spmv_csr_switches); round 6: the tree is the default
int g_row_block_dots = 0;        // liship_spmv_csr_set_row_block_dots: 1 keeps the fused dots of the dominant-pattern product on the row blocks' partial sums (the bits every other form gives)
int g_dom_march = 1;             // liship_spmv_csr_set_dom_march: 0 keeps 7-point plans with value records on the gathering dominant-pattern kernel (A/B)
int g_block_rows = 1;            // liship_spmv_csr_set_block_rows: 0 keeps plans with block rows (liship_csr_plan_encode_block_rows) on the row-by-row kernels (A/B); 2: plans of any size take them (tests)
int g_wide_union = 1;            // liship_spmv_csr_set_wide_union: 0 keeps plans whose rows take turns on several patterns off the staged value-record kernel (plan time, A/B)
int g_local_short_rows = 1;      // liship_spmv_csr_set_local_short_rows: 0 = plans of short rows never try block-local columns (rounds 2-5) -- A/B, same bits
int g_local_pairs = 1;           // liship_spmv_csr_set_local_pairs: 0 keeps the block-local kernel on one entry per lane and step (eight 2 B position loads) -- A/B, same bits
int g_reorder = 1;               // liship_spmv_csr_set_reorder: 1 (default) = the reordered form of a plan (liship_csr_plan_reorder) serves liship_csr_plan_reordered_form (whole solves), products keep
                                 // the caller's numbering; 2 = whole-matrix products of long-row plans take it too (gather of x, scattered store of y); 0 = nobody is served (A/B; the same bits)
int g_team = 1;                  // liship_spmv_csr_set_team: 0 keeps patterned rows of 8..32 entries on the one-lane-per-row pattern kernel

__device__ int d_long_row_tree = 1;   // liship_spmv_csr_set_long_row_tree (default: the tree; 0: the reference's left-to-right chain)

struct Blk { int r0, k0, r1, k1; };
// the row range of a launch and the value every row sum starts from: +0.0 as the reference's `t0 = 0.0`, or -0.0 for the
// split form, where the reference starts at the first product (`t0 = D[i]*x[i]`, lis_matvec_csr.c:70) -- (-0.0) + p is p
// for every p, signed zeros included.  Masked-out terms are added as -0.0, which leaves any sum unchanged.
struct Rows { int rb, re; double acc0; };

__global__ void csr_plan_kernel(int n, const int *__restrict__ ptr, int nblocks, int WORK, int align,
                                v2i32 *__restrict__ blk)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nblocks) return;
    const long long target = (long long)b * WORK;
    int lo = 0, hi = n;                       // first row r in [0,n] with r + ptr[r] >= target
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((long long)mid + ptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    if (align && lo < n) {
        const int al = lo & ~(ROW_ALIGN - 1);
        if (al < lo && (lo - al) + (ptr[lo] - ptr[al]) <= SLACK) lo = al;
    }
    v2i32 e; e.x = lo; e.y = ptr[lo];
    blk[b] = e;
}

__device__ __forceinline__ Blk load_blk(const v2i32 *__restrict__ blk, int b)
{
    const v2i32 lo = blk[b], hi = blk[b + 1];
    return Blk{lo.x, lo.y, hi.x, hi.y};
}

// XCD strips (round 4, structured grids).  Workgroup w runs on XCD w % 8.  In the natural order the rows that read x[j] -- its own, the neighbouring lines', the
// neighbouring planes' -- sit in workgroups on different XCDs, and x crosses the fabric once per XCD that touches it (5.2 times for the 7-point stencil at 512^3, where
// the L2 <-> fabric boundary ran at its 8 TB/s: profiles/r03_spmv512_traffic_values_streamed.json).  With strips every XCD takes one eighth of every plane of the grid --
// `plane` units of work (row blocks, 64-row workgroups ...) cover a plane, a multiple of 8 -- and walks the planes in order: the +-plane neighbours of its rows are its own
// rows of the next / previous plane, two strips apart in its own L2.  A permutation of the first (n / plane) * plane units; the tail keeps its place.
__device__ __forceinline__ int xcd_strip_unit(int w, int n, int plane)
{
    if (plane <= 0) return w;
    const int full = (n / plane) * plane;
    if (w >= full) return w;
    const int sb = plane >> 3, xcd = w & 7, slot = w >> 3;
    const int pl = slot / sb;
    return pl * plane + xcd * sb + (slot - pl * sb);
}

__device__ __forceinline__ bool clip_rows(Blk &B, const int *__restrict__ ptr, int row_begin, int row_end)
{
    if (B.r0 < row_begin) { B.r0 = row_begin; if (B.r0 < B.r1) B.k0 = ptr[B.r0]; }     // partial launches only
    if (B.r1 > row_end)   { B.r1 = row_end;   if (B.r0 < B.r1) B.k1 = ptr[B.r1]; }
    return B.r0 < B.r1;
}

// ------------------------------------------------------------------------------ products kernel
// The products of a row block are parked in LDS at GUARD + their position in the block, and the lane that owns row i adds
// its products strictly in order.  Read at the same step, the j-th products of all rows sit L slots apart for rows of L
// entries: L = 80 puts every second row on the same bank (a 25-way conflict per read).  ordered_sum therefore lets lane i
// start `skew` steps late -- step t reads the term t - skew -- with skew chosen so that (row start - skew) mod 32 = 17 i mod 32:
// at every step the lanes of a wavefront then read 32 different banks, whatever the row lengths.  Steps outside the row add
// +0.0; GUARD slots in front of the stage keep the early reads of the first rows inside the array.
constexpr int GUARD = 32;
__device__ __forceinline__ int row_skew(int start, int i) { return (start - 17 * i) & 31; }

// products for non-zeros [kbeg,kend) -> prod[GUARD + k - ka]; ka is kbeg rounded down to even
// VEC: 0 = scalar loads; 2 / 4 = 16 B value + 8 B index loads with that many independent pairs in flight per lane
// (4 for rows of 14-24 entries, 2 beyond: measured on banded and FEM patterns, tools/rowlen_sweep.py, irregular_sweep.py)
template <int BLOCK, int VEC>
__device__ __forceinline__ void stage_products(double *prod, const int *__restrict__ idx,
                                               const double *__restrict__ val,
                                               const double *__restrict__ x,
                                               int kbeg, int kend, int ka)
{
    if (VEC) {
        constexpr int BATCH = VEC ? VEC : 1;
        const int npairs = (kend - ka) >> 1;
        for (int base = 0; base < npairs; base += BATCH * BLOCK) {
            v2f64 v[BATCH];
            v2i32 c[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const int p = base + u * BLOCK + (int)threadIdx.x;
                if (p < npairs) {
                    const int k = ka + 2 * p;
                    v[u] = load_stream(reinterpret_cast<const v2f64 *>(val + k));
                    c[u] = load_stream(reinterpret_cast<const v2i32 *>(idx + k));
                }
            }
            v2f64 xv[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {       // all gathers in flight before the first use
                const int p = base + u * BLOCK + (int)threadIdx.x;
                if (p < npairs) {
                    xv[u].x = x[c[u].x]; xv[u].y = x[c[u].y];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                const int p = base + u * BLOCK + (int)threadIdx.x;
                if (p < npairs) {
                    v2f64 pr;
                    pr.x = v[u].x * xv[u].x;
                    pr.y = v[u].y * xv[u].y;
                    *reinterpret_cast<v2f64 *>(prod + GUARD + 2 * p) = pr;
                }
            }
        }
        if (((kend - ka) & 1) && threadIdx.x == BLOCK - 1) {
            const int k = kend - 1;
            prod[GUARD + k - ka] = val[k] * x[idx[k]];
        }
    } else {
        for (int k = kbeg + (int)threadIdx.x; k < kend; k += BLOCK)
            prod[GUARD + k - ka] = load_stream(val + k) * x[load_stream(idx + k)];
    }
}

// acc + q[0] + q[1] + ... + q[lim-1] with q[j] = buf[GUARD + first + j], strictly left to right (the reference's rounding
// sequence), term j read at step j + skew.  The adds form one dependent chain, so the LDS reads are issued 16 at a time ahead
// of it; absent terms are -0.0, which leaves every sum bit-unchanged (x + (-0.0) is x for all x, both zeros included).
__device__ __forceinline__ double ordered_sum(double acc, const double *buf, int first, int lim, int skew = 0)
{
    constexpr int W = 16;
    const double *p = buf + GUARD + first - skew;       // p[t] = term t - skew
    const int total = lim + skew;
    for (int t = 0; t < total; t += W) {
        double d[W];
#pragma unroll
        for (int u = 0; u < W; u++) d[u] = p[t + u];     // at most 15 slots past the row: inside the stage's slack
#pragma unroll
        for (int u = 0; u < W; u++) acc += ((unsigned)(t + u - skew) < (unsigned)lim) ? d[u] : -0.0;
    }
    return acc;
}

// the same without a skew, for the passes that continue a row longer than the stage (one lane, 10^5 terms: no selects
// in the whole batches)
__device__ __forceinline__ double ordered_sum_plain(double acc, const double *buf, int first, int lim)
{
    constexpr int W = 16;
    const double *p = buf + GUARD + first;
    int j = 0;
    for (; j + W <= lim; j += W) {
        double d[W];
#pragma unroll
        for (int u = 0; u < W; u++) d[u] = p[j + u];
#pragma unroll
        for (int u = 0; u < W; u++) acc += d[u];
    }
    if (j < lim) {
        double d[W];
#pragma unroll
        for (int u = 0; u < W; u++) d[u] = (j + u < lim) ? p[j + u] : -0.0;
#pragma unroll
        for (int u = 0; u < W; u++) acc += d[u];
    }
    return acc;
}

// fused reduction epilogue: DOT = 0 none, 1: sum_r w[r]*y[r], 2: also sum_r y[r]^2 (lane-local partial sums)
template <int DOT>
struct RowDots {
    const double *w;
    double c0, c1;
    __device__ __forceinline__ void add(int r, double yr)
    {
        if (DOT >= 1) c0 += w[r] * yr;
        if (DOT >= 2) c1 += yr * yr;
    }
    // the same with w[r] already in a register (loaded before the row's gathers, so that its latency hides behind them)
    __device__ __forceinline__ double fetch(int r) const { return DOT >= 1 ? w[r] : 0.0; }
    __device__ __forceinline__ void add_loaded(double wr, double yr)
    {
        if (DOT >= 1) c0 += wr * yr;
        if (DOT >= 2) c1 += yr * yr;
    }
};

// the same chain for the 10^5-term row, kept fed: three register batches of 7 in rotation, the reads of a batch issued two batches
// (14 dependent additions, ~120 cycles) before its terms are added, so that the additions are all that is left -- a lane that read 16,
// waited, added 16 spent half its time waiting for LDS (19.5 cycles per term against the chain's 8.7, profiles/r03_long_rows.txt).
// The loop is ONE asm statement: written in C++ the scheduler sinks the reads behind the adds and the register allocator ends every
// iteration on s_waitcnt lgkmcnt(0).  Reads run up to 14 doubles past the last whole batch (the caller leaves 32 of slack); LDS returns in
// order, so lgkmcnt(14) means "all but the two youngest batches".  Run by the owner's whole wavefront (same addresses: broadcast reads,
// every lane ends with the sum), so nothing diverges.  Same terms, same order: the reference's bits.
__device__ __forceinline__ double ordered_sum_fed(double acc, const double *buf, int first, int lim)
{
    constexpr int W = 7;
    int n = lim / (3 * W);
    const int done = n * 3 * W;
    if (n > 0) {
        unsigned ad = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) const double *)(buf + GUARD + first));
        n = __builtin_amdgcn_readfirstlane(n);
        double t[3 * W];
        asm volatile(
        "ds_read_b64 %[t0], %[ad]\n"
        "ds_read_b64 %[t1], %[ad] offset:8\n"
        "ds_read_b64 %[t2], %[ad] offset:16\n"
        "ds_read_b64 %[t3], %[ad] offset:24\n"
        "ds_read_b64 %[t4], %[ad] offset:32\n"
        "ds_read_b64 %[t5], %[ad] offset:40\n"
        "ds_read_b64 %[t6], %[ad] offset:48\n"
        "ds_read_b64 %[t7], %[ad] offset:56\n"
        "ds_read_b64 %[t8], %[ad] offset:64\n"
        "ds_read_b64 %[t9], %[ad] offset:72\n"
        "ds_read_b64 %[t10], %[ad] offset:80\n"
        "ds_read_b64 %[t11], %[ad] offset:88\n"
        "ds_read_b64 %[t12], %[ad] offset:96\n"
        "ds_read_b64 %[t13], %[ad] offset:104\n"
        "1:\n"
        "ds_read_b64 %[t14], %[ad] offset:112\n"
        "ds_read_b64 %[t15], %[ad] offset:120\n"
        "ds_read_b64 %[t16], %[ad] offset:128\n"
        "ds_read_b64 %[t17], %[ad] offset:136\n"
        "ds_read_b64 %[t18], %[ad] offset:144\n"
        "ds_read_b64 %[t19], %[ad] offset:152\n"
        "ds_read_b64 %[t20], %[ad] offset:160\n"
        "s_waitcnt lgkmcnt(14)\n"
        "v_add_f64 %[acc], %[acc], %[t0]\n"
        "v_add_f64 %[acc], %[acc], %[t1]\n"
        "v_add_f64 %[acc], %[acc], %[t2]\n"
        "v_add_f64 %[acc], %[acc], %[t3]\n"
        "v_add_f64 %[acc], %[acc], %[t4]\n"
        "v_add_f64 %[acc], %[acc], %[t5]\n"
        "v_add_f64 %[acc], %[acc], %[t6]\n"
        "ds_read_b64 %[t0], %[ad] offset:168\n"
        "ds_read_b64 %[t1], %[ad] offset:176\n"
        "ds_read_b64 %[t2], %[ad] offset:184\n"
        "ds_read_b64 %[t3], %[ad] offset:192\n"
        "ds_read_b64 %[t4], %[ad] offset:200\n"
        "ds_read_b64 %[t5], %[ad] offset:208\n"
        "ds_read_b64 %[t6], %[ad] offset:216\n"
        "s_waitcnt lgkmcnt(14)\n"
        "v_add_f64 %[acc], %[acc], %[t7]\n"
        "v_add_f64 %[acc], %[acc], %[t8]\n"
        "v_add_f64 %[acc], %[acc], %[t9]\n"
        "v_add_f64 %[acc], %[acc], %[t10]\n"
        "v_add_f64 %[acc], %[acc], %[t11]\n"
        "v_add_f64 %[acc], %[acc], %[t12]\n"
        "v_add_f64 %[acc], %[acc], %[t13]\n"
        "ds_read_b64 %[t7], %[ad] offset:224\n"
        "ds_read_b64 %[t8], %[ad] offset:232\n"
        "ds_read_b64 %[t9], %[ad] offset:240\n"
        "ds_read_b64 %[t10], %[ad] offset:248\n"
        "ds_read_b64 %[t11], %[ad] offset:256\n"
        "ds_read_b64 %[t12], %[ad] offset:264\n"
        "ds_read_b64 %[t13], %[ad] offset:272\n"
        "s_waitcnt lgkmcnt(14)\n"
        "v_add_f64 %[acc], %[acc], %[t14]\n"
        "v_add_f64 %[acc], %[acc], %[t15]\n"
        "v_add_f64 %[acc], %[acc], %[t16]\n"
        "v_add_f64 %[acc], %[acc], %[t17]\n"
        "v_add_f64 %[acc], %[acc], %[t18]\n"
        "v_add_f64 %[acc], %[acc], %[t19]\n"
        "v_add_f64 %[acc], %[acc], %[t20]\n"
        "v_add_u32 %[ad], 168, %[ad]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [acc] "+v"(acc), [ad] "+v"(ad), [n] "+s"(n), [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [t4] "=&v"(t[4]), [t5] "=&v"(t[5]), [t6] "=&v"(t[6]), [t7] "=&v"(t[7]), [t8] "=&v"(t[8]), [t9] "=&v"(t[9]), [t10] "=&v"(t[10]), [t11] "=&v"(t[11]), [t12] "=&v"(t[12]), [t13] "=&v"(t[13]), [t14] "=&v"(t[14]), [t15] "=&v"(t[15]), [t16] "=&v"(t[16]), [t17] "=&v"(t[17]), [t18] "=&v"(t[18]), [t19] "=&v"(t[19]), [t20] "=&v"(t[20])
        :
        : "scc", "memory");
    }
    return ordered_sum_plain(acc, buf, first + done, lim - done);
}

// The rows of a wavefront when they all have ONE length L (the interior rows of a finite-element or stencil matrix): no lane needs a
// mask, an odd L (or 2 x odd: two-way) spreads the lanes' reads over the banks without a skew -- which costs every wavefront up to 31 more
// steps --, and the chain can be kept fed as the long row's is: L mod 21 terms first (all read at once, a uniform jump into the run of
// additions), then whole rounds of 21 through ordered_sum_fed's loop, each lane from its own address.  Same terms, same order.
__device__ __forceinline__ double ordered_sum_first20(double acc, const double *buf, int first, int rem)     // rem uniform, 0..20
{
    const double *p = buf + GUARD + first + rem - 20;     // p[u] = term rem - 20 + u (u < 20 - rem: slots in front of the row, read and dropped)
    double d[20];
#pragma unroll
    for (int u = 0; u < 20; u++) d[u] = p[u];
    switch (rem) {
    case 20: acc += d[0]; [[fallthrough]];
    case 19: acc += d[1]; [[fallthrough]];
    case 18: acc += d[2]; [[fallthrough]];
    case 17: acc += d[3]; [[fallthrough]];
    case 16: acc += d[4]; [[fallthrough]];
    case 15: acc += d[5]; [[fallthrough]];
    case 14: acc += d[6]; [[fallthrough]];
    case 13: acc += d[7]; [[fallthrough]];
    case 12: acc += d[8]; [[fallthrough]];
    case 11: acc += d[9]; [[fallthrough]];
    case 10: acc += d[10]; [[fallthrough]];
    case 9: acc += d[11]; [[fallthrough]];
    case 8: acc += d[12]; [[fallthrough]];
    case 7: acc += d[13]; [[fallthrough]];
    case 6: acc += d[14]; [[fallthrough]];
    case 5: acc += d[15]; [[fallthrough]];
    case 4: acc += d[16]; [[fallthrough]];
    case 3: acc += d[17]; [[fallthrough]];
    case 2: acc += d[18]; [[fallthrough]];
    case 1: acc += d[19]; [[fallthrough]];
    default: break;
    }
    return acc;
}

// the sums of the rows a wavefront's lanes own (first, lim: per lane), whichever way fits
__device__ __forceinline__ double ordered_sum_rows(double acc, const double *buf, int first, int lim, int lane, int uniform)
{
    const int L0 = __builtin_amdgcn_readfirstlane(lim);
    if (uniform != 0 && L0 >= 21 && (L0 & 3) != 0 && __builtin_amdgcn_ballot_w64(lim != L0) == 0) {      // (uniform)
        const int rem = L0 % 21;
        acc = ordered_sum_first20(acc, buf, first, rem);
        return ordered_sum_fed(acc, buf, first + rem, L0 - rem);
    }
    return ordered_sum(acc, buf, first, lim, row_skew(first, lane));
}

// one whole row block of any shape (many empty rows, rows longer than the LDS stage).
// Invariant from the plan: every row but the last ends inside the first pass of CAP products.
// ------------------------------------------------------------------------------ long rows in tree mode: the tail over many workgroups
// (The default since round 6; LIS_AMD_LONG_ROW_CHAIN=1 / liship_spmv_csr_set_long_row_tree(0) switch it off.)  One workgroup folding a 200 000-entry row is ~100 dependent rounds of memory latency: the heavy-tailed stress matrix ran at 0.78 ms with the
// workgroup tree where its bytes are worth 0.09 ms.  So the part of a row block beyond its first TAIL_FROM entries -- always the tail of its last row -- is cut into chunks of
// TAIL_CHUNK entries (the plan lists them: build_split), a workgroup per chunk adds its products (lane sums, wavefront butterfly, wavefronts in order: a fixed order),
// spmv_csr_tail_fold_kernel adds a block's chunk sums in chunk order and leaves the total in y[that row], and block_by_products -- which every kernel reaches for such a
// block -- adds it to the head of the row it has summed itself.  Both passes run in front of every launch of a plan that has such blocks (tail_prepass, spmv_csr.hip) with
// the launch's own row range, so they see the block as the product's kernel will.
constexpr int TAIL_FROM = 16384, TAIL_CHUNK = 8192, TREE_MIN = 1024;
template <int BLOCK>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_tail_chunks_kernel(const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x,
                                 const v2i32 *__restrict__ blk, const v2i32 *__restrict__ tchunk, Rows RW, double *__restrict__ tpart)
{
    __shared__ double scratch[BLOCK / WAVE];
    const v2i32 c = tchunk[blockIdx.x];                 // {row block, chunk of its tail}
    Blk B = load_blk(blk, c.x);
    double part = 0.0;
    if (clip_rows(B, ptr, RW.rb, RW.re) && B.k1 - B.k0 > TAIL_FROM) {
        const int kb = B.k0 + TAIL_FROM + c.y * TAIL_CHUNK, ke = min(B.k1, kb + TAIL_CHUNK);
        for (int k = kb + (int)threadIdx.x; k < ke; k += 8 * BLOCK) {
            double v[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int kk = min(k + u * BLOCK, ke - 1); v[u] = load_stream(val + kk); xv[u] = x[load_stream(idx + kk)]; }
#pragma unroll
            for (int u = 0; u < 8; u++) part += (k + u * BLOCK < ke) ? v[u] * xv[u] : 0.0;
        }
    }
    const double tot = block_sum<BLOCK / WAVE>(part, scratch);
    if (threadIdx.x == 0) tpart[blockIdx.x] = tot;
}
__global__ void spmv_csr_tail_fold_kernel(int nheavy, const int *__restrict__ ptr, const v2i32 *__restrict__ blk, const v2i32 *__restrict__ thead, Rows RW,
                                          const double *__restrict__ tpart, double *__restrict__ y, const int *__restrict__ rowmap)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nheavy) return;
    const v2i32 hd = thead[h], nx = thead[h + 1];       // {row block, its first chunk}
    Blk B = load_blk(blk, hd.x);
    if (!clip_rows(B, ptr, RW.rb, RW.re) || B.k1 - B.k0 <= TAIL_FROM) return;
    double s = 0.0;
    for (int c = hd.y; c < nx.y; c++) s += tpart[c];
    const int rl = B.r1 - 1;
    y[rowmap ? rowmap[rl] : rl] = s;
}

template <int BLOCK, int CAP, int VEC, int DOT = 0>
__device__ __forceinline__ void block_by_products(double *prod, const int *__restrict__ ptr,
                                                  const int *__restrict__ idx, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y,
                                                  const Blk B, RowDots<DOT> &dots, const double acc0 = 0.0,
                                                  const int *__restrict__ rowmap = nullptr)      // (reordered plans: row r of this matrix is y[rowmap[r]])
{
    const int r0 = B.r0, r1 = B.r1, k0 = B.k0, k1 = B.k1;
    const int ka = k0 & ~1;
    const int kfirst = min(k1, k0 + CAP);      // end of the first pass

    // this lane's first row extent: issued before the streaming loads so it is back by the sums
    const int rmine = r0 + (int)threadIdx.x;
    int s_first = 0, e_first = 0;
    if (rmine < r1) { s_first = ptr[rmine]; e_first = ptr[rmine + 1]; }

    stage_products<BLOCK, VEC>(prod, idx, val, x, k0, kfirst, ka);
    __syncthreads();

    double carry = 0.0;
    for (int r = rmine; r < r1; r += BLOCK) {
        int s = s_first, e = e_first;
        if (r != rmine) { s = ptr[r]; e = ptr[r + 1]; }
        const double acc = ordered_sum_rows(acc0, prod, s - ka, min(e, kfirst) - s, (int)threadIdx.x, 1);
        if (e <= kfirst) { store_stream(y + (rowmap ? rowmap[r] : r), acc); dots.add(r, acc); } else carry = acc;   // only the block's last row can overflow
    }

    if (k1 > kfirst) {                                          // uniform: finish the long last row
        const int rl = r1 - 1;
        const int owner = (rl - r0) % BLOCK;
        // the tree (the default: liship_spmv_csr_set_long_row_tree) only where there is a chain worth cutting: at least TREE_MIN entries beyond the first pass.  A row
        // block's last row may straddle the end of the stage whatever its length (81-entry FEM rows in the block-local kernel's blocks do): those few dozen entries stay
        // on the ordered chain, so every row SHORTER than TREE_MIN entries keeps the reference's bits in every kernel, tree or not
        const bool tree = d_long_row_tree != 0 && k1 - kfirst >= TREE_MIN;
        __shared__ double tree_scratch[BLOCK / WAVE];
        int base = kfirst;
        if (tree) {
            // every lane adds the products it reaches with stride BLOCK (8 loads and gathers in flight, nothing staged), the
            // workgroup folds the lane sums (butterfly, then the wavefronts in order): a fixed order, but not left to right --
            // a 200 000-entry row costs ~100 rounds of memory latency instead of a 200 000-long dependent add chain
            // beyond TAIL_FROM entries the tail was summed by the launch's prepass (a workgroup per TAIL_CHUNK entries) and waits in y[this row]
            static_assert(CAP < TAIL_FROM, "the prepass starts beyond the first pass's stage");
            const bool heavy = k1 - k0 > TAIL_FROM;
            const int kt = heavy ? k0 + TAIL_FROM : k1;
            double part = 0.0;
            for (int k = kfirst + (int)threadIdx.x; k < kt; k += 8 * BLOCK) {
                double v[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int kk = min(k + u * BLOCK, kt - 1); v[u] = load_stream(val + kk); xv[u] = x[load_stream(idx + kk)]; }
#pragma unroll
                for (int u = 0; u < 8; u++) part += (k + u * BLOCK < kt) ? v[u] * xv[u] : 0.0;
            }
            __syncthreads();
            const double tot = block_sum<BLOCK / WAVE>(part, tree_scratch);
            if (threadIdx.x == 0) tree_scratch[0] = tot;
            __syncthreads();
            if ((int)threadIdx.x == owner) { carry += tree_scratch[0]; if (heavy) carry += y[rowmap ? rowmap[rl] : rl]; }
            base = k1;
        }
        static_assert(BLOCK > WAVE, "the owner of a long row is fed by the lanes of the OTHER wavefronts");
        if (base < k1) {
            // The rest of the long row, strictly in order: ONE lane owns the chain of additions, and nothing can shorten it but
            // keeping that lane fed.  The stage is used as two halves: while the owner adds the products of one half, the lanes
            // of the OTHER wavefronts form the products of the next half (all of it in flight at once), so a pass costs
            // the longer of the two instead of their sum, and the owner's wavefront never waits for memory.
            constexpr int helpers = BLOCK - WAVE;                      // lanes of the other wavefronts
            constexpr int HFIT = ((CAP - 8) / 2) & ~15;         // products per half that fit (32 doubles of slack behind the second: ordered_sum_fed reads ahead)
            constexpr int HCAP = (6 * helpers) & ~15;                  // ... and at most six per helper lane (registers)
            constexpr int H = HFIT < HCAP ? HFIT : HCAP;
            const int ownerwave = owner / WAVE;
            const int hl = ((int)threadIdx.x / WAVE < ownerwave) ? (int)threadIdx.x : (int)threadIdx.x - WAVE;   // index among them
            const bool helper = (int)threadIdx.x / WAVE != ownerwave;
            // A helper lane takes PER entries of a half -- the whole half is in flight at once -- and the loads run TWO steps ahead of the
            // owner: the values and x of half s+2 are issued (into registers) before the barrier that ends step s and become products in
            // LDS at the start of step s+1, their column indices were fetched a step before that.  A round trip under the random gathers of
            // the rest of the matrix (4-5 us) has a whole step of the owner (3 us of additions) to come back.
            constexpr int PER = (H + helpers - 1) / helpers;
            int nidx[PER];
            double v[PER], xv[PER];
            auto fetch_indices = [&](int kb) {                  // indices of the half starting at kb
                const int ke = min(kb + H, k1);
#pragma unroll
                for (int u = 0; u < PER; u++) nidx[u] = (kb < ke) ? load_stream(idx + min(kb + hl + u * helpers, ke - 1)) : 0;
            };
            auto issue_loads = [&](int kb) {                    // values and x of the half starting at kb (its indices are in nidx)
                const int ke = min(kb + H, k1);
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    v[u] = kb < ke ? load_stream(val + min(kb + hl + u * helpers, ke - 1)) : 0.0;
                    xv[u] = (kb >= ke) ? 1.0 : x[nidx[u]];
                }
            };
            auto write_products = [&](int half, int kb) {       // prod[GUARD + half * H + (k - kb)] = value[k] * x[index[k]]
                const int ke = min(kb + H, k1);
                double *dst = prod + GUARD + half * H;
#pragma unroll
                for (int u = 0; u < PER; u++) if (kb + hl + u * helpers < ke) dst[hl + u * helpers] = v[u] * xv[u];
            };
            __syncthreads();                                    // the first pass's sums have read the stage
            // two loops with the same barriers, one per role (the branch is uniform per wavefront): in one loop the registers of the loads in
            // flight and those of the chain would be live together in every kernel that inlines this path
            if (helper) {
                fetch_indices(base); issue_loads(base); fetch_indices(base + H);
                write_products(0, base);
                issue_loads(base + H); fetch_indices(base + 2 * H);
                __syncthreads();
                for (int half = 0, kb = base; kb < k1; kb += H, half ^= 1) {
                    write_products(half ^ 1, kb + H);           // loaded during the step before
                    issue_loads(kb + 2 * H); fetch_indices(kb + 3 * H);
                    __syncthreads();
                }
            } else {                                            // the owner's wavefront adds as one: every lane starts from the owner's sum
                const int ol = __builtin_amdgcn_readfirstlane(owner & (WAVE - 1));
                carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(carry), ol), __builtin_amdgcn_readlane(__double2loint(carry), ol));
                __syncthreads();
                for (int half = 0, kb = base; kb < k1; kb += H, half ^= 1) {
                    carry = ordered_sum_fed(carry, prod, half * H, min(kb + H, k1) - kb);
                    __syncthreads();
                }
            }
        }
        if ((int)threadIdx.x == owner) { store_stream(y + (rowmap ? rowmap[rl] : rl), carry); dots.add(rl, carry); }
    }
}

// lane partials -> one value per workgroup (fixed order: wave butterfly, then waves in order), written by lane 0
template <int BLOCK, int DOT>
__device__ __forceinline__ void publish_dots(const RowDots<DOT> &dots, double *scratch, double *partial, int slot, int stride)
{
    if (DOT == 0) return;
    const double t0 = block_sum<BLOCK / WAVE>(dots.c0, scratch);
    if (threadIdx.x == 0) partial[slot] = t0;
    if (DOT >= 2) {
        const double t1 = block_sum<BLOCK / WAVE>(dots.c1, scratch);
        if (threadIdx.x == 0) partial[stride + slot] = t1;
    }
}

template <int BLOCK, int WORK, int VEC, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_products_kernel(const int *__restrict__ ptr, const int *__restrict__ idx,
                              const double *__restrict__ val, const double *__restrict__ x,
                              double *__restrict__ y, const v2i32 *__restrict__ blk,
                              int bfirst, int nb, Rows RW,
                              const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                              const double *__restrict__ guard = nullptr, int pstride = 0, const int *__restrict__ order = nullptr)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    constexpr int CAP = WORK + SLACK;
    __shared__ __attribute__((aligned(16))) double prod[(GUARD + CAP + 8 + 16)];
    __shared__ double dot_scratch[BLOCK / WAVE];
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    // order: the plan's launch order when some row blocks hold a row far longer than the stage -- those first, so that their
    // chains of additions (one lane, strictly in order) run beside the rest of the matrix instead of behind it
    const int lb = order ? order[blockIdx.x] : (int)blockIdx.x;
    Blk B = load_blk(blk, bfirst + lb);
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    block_by_products<BLOCK, CAP, VEC, DOT>(prod, ptr, idx, val, x, y, B, dots, acc0);
    __syncthreads();
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// ------------------------------------------------------------------------------ row-gather kernel
// xs_plane: XCD strips (xcd_strip_unit) -- row blocks per plane of a structured grid, or 0 for the natural order.  A permutation of the launch's blocks: the
// partial sums of the fused dots stay in block order.
// (Round 5, measured and dropped: the index slice issued ahead of the value slice, one bare barrier once every wavefront's index pieces have landed, and the x gathers
//  of a row's first U entries issued behind the value stream instead of after it -- everything by hand, the compiler waits with vmcnt(0) for loads whose age it cannot
//  count.  512^3, same box, interleaved: 2.3839 against 2.3843 ms at 256 / 2048, 2.421 against 2.411 at 192 / 1408.  The fabric binds this kernel, not its prologue.)
template <int BLOCK, int WORK, int U, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_rowgather_kernel(const int *__restrict__ ptr, const int *__restrict__ idx,
                               const double *__restrict__ val, const double *__restrict__ x,
                               double *__restrict__ y, const v2i32 *__restrict__ blk,
                               int bfirst, int nb, Rows RW, int nnz_total, int xs_plane,
                               const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                               const double *__restrict__ guard = nullptr, int pstride = 0,
                               const int *__restrict__ rowmap = nullptr)      // reordered plans: row r of the matrix this launch walks is y[rowmap[r]]
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    __shared__ double dot_scratch[BLOCK / WAVE];
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    constexpr int CAP = WORK + SLACK;
    // + one wavefront of slack: the LDS-DMA form always lands whole 1 KiB wave slices
    __shared__ __attribute__((aligned(16))) double valL[(GUARD + CAP + 8 + 16) + 2 * WAVE];    // also the padded product stage of block_by_products
    __shared__ __attribute__((aligned(16))) int idxL[CAP + 8 + 4 * WAVE];

    const int lb = xcd_strip_unit((int)blockIdx.x, nb, xs_plane);
    Blk B = load_blk(blk, bfirst + lb);
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }

    const int ka = B.k0 & ~3;                       // 16 B aligned start for both streams
    const int nq = (B.k1 - ka + 3) >> 2;            // quads of 4 non-zeros
    if ((B.k1 - ka) > CAP || ka + 4 * nq > nnz_total) {     // long row / tail of the arrays
        block_by_products<BLOCK, CAP, 4, DOT>(valL, ptr, idx, val, x, y, B, dots, acc0, rowmap);
        __syncthreads();
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }

    const int rmine = B.r0 + (int)threadIdx.x;
    int s_first = 0, e_first = 0, y_first = rmine;
    if (rmine < B.r1) { s_first = ptr[rmine]; e_first = ptr[rmine + 1]; if (rowmap) y_first = rowmap[rmine]; }

    // linear copies of both slices, every load instruction fully coalesced (16 B per lane, 1 KiB per wave): global -> LDS directly, each wave lands
    // 64 x 16 B at a wave-uniform LDS base; lanes past the end re-read the last valid 16 B (their LDS slot is never used)
    const int np = 2 * nq;                          // pairs of values
    {
        const int wbase = (int)threadIdx.x & ~(WAVE - 1), lane = (int)threadIdx.x & (WAVE - 1);
        for (int p0 = wbase; p0 < np; p0 += BLOCK) {
            const int p = min(p0 + lane, np - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
        }
        for (int q0 = wbase; q0 < nq; q0 += BLOCK) {
            const int q = min(q0 + lane, nq - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v4i32 *>(idx + ka) + q),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v4i32 *>(idxL) + q0), 16, 0, 2);
        }
    }
    __syncthreads();

    for (int r = rmine; r < B.r1; r += BLOCK) {
        int s = s_first, e = e_first, yr = y_first;
        if (r != rmine) { s = ptr[r]; e = ptr[r + 1]; yr = rowmap ? rowmap[r] : r; }
        const int len = e - s;
        const int off = s - ka;
        const double wr = dots.fetch(r);
        double acc = acc0;
        for (int j0 = 0; j0 < len; j0 += U) {
            int cc[U]; double vv[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = min(j0 + u, len - 1);         // clamped: repeats the row's last entry
                cc[u] = idxL[off + j];
                vv[u] = valL[off + j];
            }
#pragma unroll
            for (int u = 0; u < U; u++) xx[u] = x[cc[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double t = vv[u] * xx[u];
                acc += (j0 + u < len) ? t : -0.0;           // -0.0 terms leave any sum bit-unchanged
            }
        }
        store_stream(y + yr, acc);
        dots.add_loaded(wr, acc);
    }
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// ------------------------------------------------------------------------------ products kernel, block-local columns
// Long rows are bound by the x gather, not by the streams (profiles/r02_csr_kernel_experiments.txt: without the gather
// the products kernel runs at 80 % of the roofline, with it at 62 %).  But the entries of one row block address few
// DISTINCT columns when rows are long: the three rows of a finite-element node share all their columns, neighbouring
// rows most of them -- 2048 entries of the 81-per-row pattern touch ~270 columns.  So the plan can keep, per row block,
// the sorted list of its distinct columns (dcol, 4 B each) and per non-zero a 2 B position in that list (lcol,
// liship_csr_plan_localize_columns), and this kernel
//   * loads the block's list and gathers x ONCE per distinct column into LDS (4 per lane, sorted columns: neighbouring
//     lanes hit the same lines), while the value / position slices arrive by LDS-DMA,
//   * forms every product from LDS alone (all lanes, consecutive entries),
//   * adds the products of a row strictly left to right, one lane per row, as block_by_products does.
// 10 B + 4 B x (distinct / entries) per non-zero instead of 12, an eighth of the gathers; same terms in the same
// order, so y is bit-identical.  Row blocks the plan left out (a row longer than the stage, more than 4 x BLOCK
// distinct columns) have an empty list and take block_by_products on the 4 B indices.
// RPOS (round 4): the 2 B positions go from HBM to REGISTERS (a lane only ever reads the positions of its own eight items), not through LDS; XCAP: the
// longest list the plan found, rounded up to 1024 / 1536 / 2048 -- the x stage in LDS is no larger than that.
// RUNS (round 5): the sorted distinct columns of a row block of a mesh with 3 unknowns per node come in TRIPLES of consecutive columns; when every list of the plan
// is made of such triples (liship_csr_plan_localize_columns checks) the kernel reads one 4 B run start per triple instead of three 4 B columns -- a third of the list
// bytes (Queen class: 384 -> 128 MB of 3.34 GB) and a third of the list loads -- and a lane gathers its triple's three x.  Same stage contents: same bits.
template <int BLOCK, int WORK, int DOT = 0, int NDPL = 2, int XCAP = NDPL * BLOCK, bool RPOS = false, bool RUNS = false, bool PAIR = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((RPOS && XCAP <= 1536) ? 8 : 4)))      // four 512-lane workgroups per CU want <= 64 VGPRs (the fused-dot forms took 66 - 68)
void spmv_csr_local_kernel(const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                           const unsigned short *__restrict__ lcol, const int *__restrict__ dcol,
                           const int *__restrict__ doff, const double *__restrict__ x, double *__restrict__ y,
                           const v2i32 *__restrict__ blk, int bfirst, int nb, Rows RW, int nnz_total,
                           const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                           const double *__restrict__ guard = nullptr, int pstride = 0, int uniform = 1,
                           const int *__restrict__ drun = nullptr, const int *__restrict__ droff = nullptr,
                           const int *__restrict__ rowmap = nullptr)      // reordered plans (liship_csr_plan_reorder): row r of the matrix this launch walks is y[rowmap[r]]
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    constexpr int CAP = WORK + SLACK + 8;           // + the 8-entry alignment of the position slice
    static_assert(XCAP <= NDPL * BLOCK && XCAP % 4 == 0, "the list is loaded NDPL columns per lane");
    static_assert(!RPOS || CAP <= 8 * BLOCK, "eight items per lane");
    __shared__ double dot_scratch[BLOCK / WAVE];
    __shared__ __attribute__((aligned(16))) double valL[(GUARD + CAP + 8 + 16) + 2 * WAVE];
    __shared__ __attribute__((aligned(16))) unsigned short lcL[RPOS ? 8 : CAP + 8 + 8 * WAVE];
    __shared__ __attribute__((aligned(16))) double xL[XCAP];
    RowDots<DOT> dots{wdot, 0.0, 0.0};

    const int lb = blockIdx.x;
    Blk B = load_blk(blk, bfirst + lb);
    const int d0 = doff[bfirst + lb], nd = doff[bfirst + lb + 1] - d0;      // the plan's block, whatever the row range of this launch
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    const int ka = B.k0 & ~7;                       // 16 B aligned start of the position slice (64 B for the values)
    const int cnt = B.k1 - ka;
    const int np = (cnt + 1) >> 1;                  // 16 B pieces of the value slice
    const int nl = (cnt + 7) >> 3;                  // 16 B pieces of the position slice (the array is padded)
    if (nd == 0 || nd > XCAP || cnt > CAP || ka + 2 * np > nnz_total) {  // block without a list / last value of the array
        block_by_products<BLOCK, CAP, 4, DOT>(valL, ptr, idx, val, x, y, B, dots, acc0, rowmap);
        __syncthreads();
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    // the block's distinct columns: the first memory operation, so that the gathers can leave while the slices stream in
    const int j4 = NDPL * (int)threadIdx.x;
    v4i32 dc = {0, 0, 0, 0};
    constexpr int NRPL = (XCAP / 3 + BLOCK - 1) / BLOCK;      // (RUNS) triples per lane
    int rs[NRPL], nr = 0;
    if (RUNS) {
        const int q0 = droff[bfirst + lb];
        nr = droff[bfirst + lb + 1] - q0;
#pragma unroll
        for (int k = 0; k < NRPL; k++) { const int j = k * BLOCK + (int)threadIdx.x; rs[k] = j < nr ? drun[q0 + j] : -1; }
    } else if (j4 < nd) {                           // lists are padded to whole 16 B pieces
        if (NDPL == 4) dc = *reinterpret_cast<const v4i32 *>(dcol + d0 + j4);
        else { const v2i32 h = *reinterpret_cast<const v2i32 *>(dcol + d0 + j4); dc.x = h.x; dc.y = h.y; }
    }
    {
        const int wbase = (int)threadIdx.x & ~(WAVE - 1), lane = (int)threadIdx.x & (WAVE - 1);
        for (int p0 = wbase; p0 < np; p0 += BLOCK) {
            const int p = min(p0 + lane, np - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL + GUARD) + p0), 16, 0, 2);
        }
        if (!RPOS) for (int q0 = wbase; q0 < nl; q0 += BLOCK) {
            const int q = min(q0 + lane, nl - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v4i32 *>(lcol + ka) + q),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v4i32 *>(lcL) + q0), 16, 0, 2);
        }
    }
    // RPOS: the positions of this lane's own items (entry B.k0 + lane + g * BLOCK of the matrix), eight 2 B loads -- a wavefront's 64 lanes read 128 B each
    // PAIR: a lane owns PAIRS of neighbouring entries (2 (lane + g BLOCK), + 1 counted from the aligned start ka): four 4 B position loads instead of eight 2 B ones,
    // 16 B LDS accesses in the products phase.  The up to 7 entries in front of the block's first one come along (the previous block's values and positions: a product
    // nobody reads, parked in front of the first row)
    unsigned short pos[8];
    unsigned pp[4];
    const int cnt2 = (cnt + 1) & ~1;
    if (RPOS && PAIR) {
#pragma unroll
        for (int g = 0; g < 4; g++) pp[g] = *reinterpret_cast<const unsigned *>(lcol + ka + min(2 * ((int)threadIdx.x + g * BLOCK), cnt2 - 2));
    } else if (RPOS) {
#pragma unroll
        for (int g = 0; g < 8; g++) pos[g] = lcol[min(B.k0 + (int)threadIdx.x + g * BLOCK, B.k1 - 1)];
    }
    // rows stay with consecutive lanes of as few wavefronts as possible: dealing them to all wavefronts was tried and multiplies
    // the LDS instructions of the serial sums by the number of wavefronts (each then issues the whole chain for a few lanes)
    const int rmine = B.r0 + (int)threadIdx.x;
    int s_first = 0, e_first = 0, y_first = rmine;
    if (rmine < B.r1) { s_first = ptr[rmine]; e_first = ptr[rmine + 1]; if (rowmap) y_first = rowmap[rmine]; }
    if (RUNS) {
        double t0[NRPL], t1[NRPL], t2[NRPL];
#pragma unroll
        for (int k = 0; k < NRPL; k++) if (rs[k] >= 0) { t0[k] = x[rs[k]]; t1[k] = x[rs[k] + 1]; t2[k] = x[rs[k] + 2]; }
#pragma unroll
        for (int k = 0; k < NRPL; k++) if (rs[k] >= 0) { double *q = xL + 3 * (k * BLOCK + (int)threadIdx.x); q[0] = t0[k]; q[1] = t1[k]; q[2] = t2[k]; }
    } else if (j4 < nd) {
        v2f64 a, b;
        a.x = x[dc.x]; a.y = x[dc.y];
        if (NDPL == 4) { b.x = x[dc.z]; b.y = x[dc.w]; }
        *reinterpret_cast<v2f64 *>(xL + j4) = a;
        if (NDPL == 4) *reinterpret_cast<v2f64 *>(xL + j4 + 2) = b;
    }
    __syncthreads();

    // products in place, from LDS alone: entry e of the stage by lane e % BLOCK, 8 in flight per lane
    if (RPOS && PAIR) {
        v2f64 vv[4];
        double x0[4], x1[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int e = min(2 * ((int)threadIdx.x + g * BLOCK), cnt2 - 2);
            vv[g] = *reinterpret_cast<const v2f64 *>(valL + GUARD + e);
            x0[g] = xL[min((int)(pp[g] & 0xffffu), XCAP - 1)]; x1[g] = xL[min((int)(pp[g] >> 16), XCAP - 1)];
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int e = 2 * ((int)threadIdx.x + g * BLOCK);
            if (e < cnt) { v2f64 pr; pr.x = vv[g].x * x0[g]; pr.y = vv[g].y * x1[g]; *reinterpret_cast<v2f64 *>(valL + GUARD + e) = pr; }
        }
    } else if (RPOS) {
        const int e0 = (B.k0 - ka) + (int)threadIdx.x;
        double vv[8], xv[8];
#pragma unroll
        for (int g = 0; g < 8; g++) { const int e = min(e0 + g * BLOCK, cnt - 1); vv[g] = valL[GUARD + e]; xv[g] = xL[pos[g]]; }
#pragma unroll
        for (int g = 0; g < 8; g++) { const int e = e0 + g * BLOCK; if (e < cnt) valL[GUARD + e] = vv[g] * xv[g]; }
    } else
    for (int e0 = (B.k0 - ka) + (int)threadIdx.x; e0 < cnt; e0 += 8 * BLOCK) {
        double vv[8], xv[8];
#pragma unroll
        for (int g = 0; g < 8; g++) { const int e = min(e0 + g * BLOCK, cnt - 1); vv[g] = valL[GUARD + e]; xv[g] = xL[lcL[e]]; }
#pragma unroll
        for (int g = 0; g < 8; g++) { const int e = e0 + g * BLOCK; if (e < cnt) valL[GUARD + e] = vv[g] * xv[g]; }
    }
    __syncthreads();

    for (int r = rmine; r < B.r1; r += BLOCK) {
        int s = s_first, e = e_first, yr = y_first;
        if (r != rmine) { s = ptr[r]; e = ptr[r + 1]; yr = rowmap ? rowmap[r] : r; }
        const double wr = dots.fetch(r);
        const double acc = ordered_sum_rows(acc0, valL, s - ka, e - s, (int)threadIdx.x, uniform);
        store_stream(y + yr, acc);
        dots.add_loaded(wr, acc);
    }
    if (DOT != 0) __syncthreads();
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// x in the numbering of a reordered plan: xp[i] = x[perm[i]], THREE consecutive entries per lane -- the walk keeps the unknowns of a node together, so on a 3-dof mesh a
// lane's three reads are one 24 B piece of x; any other permutation is served all the same.  Bound by the 128 B lines those pieces arrive in -- 1.37 M of them, 0.030 ms
// on the Queen-class matrix, whether x is cache-resident or not and whatever the cache-policy bits of the loads (tools/permute_probe.py, profiles/EXPERIMENTS.md)
__global__ __launch_bounds__(256)
void csr_reorder_gather_kernel(int n, const int *__restrict__ perm, const double *__restrict__ x, double *__restrict__ xp)
{
    const int i = 3 * (blockIdx.x * 256 + (int)threadIdx.x);
    if (i + 2 < n) {
        const int a = perm[i], b = perm[i + 1], c = perm[i + 2];
        const double va = x[a], vb = x[b], vc = x[c];
        xp[i] = va; xp[i + 1] = vb; xp[i + 2] = vc;
    } else for (int k = i; k < n; k++) xp[k] = x[perm[k]];
}

// and back: x[perm[i]] = xp[i] (24 B pieces on a 3-dof mesh)
__global__ __launch_bounds__(256)
void csr_reorder_scatter_kernel(int n, const int *__restrict__ perm, const double *__restrict__ xp, double *__restrict__ x)
{
    const int i = 3 * (blockIdx.x * 256 + (int)threadIdx.x);
    if (i + 2 < n) {
        const int a = perm[i], b = perm[i + 1], c = perm[i + 2];
        const double va = xp[i], vb = xp[i + 1], vc = xp[i + 2];
        x[a] = va; x[b] = vb; x[c] = vc;
    } else for (int k = i; k < n; k++) x[perm[k]] = xp[k];
}

// plan time, one workgroup per row block: sort the block's column indices (bitonic, in LDS), keep the distinct ones.
// PASS 0 counts them (nd[b]; 0 = no list: empty block, a row longer than the stage, more than ndmax columns);
// PASS 1 writes the list at dcol[doff[b]...) (padded to a multiple of 4 with its last entry) and, per non-zero, the
// position of its column in the list.
// LOCAL_SORT: the power of two the bitonic network sorts (>= the block's entries), DCAP: room for its distinct columns.  (8192, 4352) serves the 4096-item blocks
// of rounds 2-3; the 3584- / 3072-item blocks of the register-position kernels fit (4096, 3840): 78 stages over 4096 keys instead of 91 over 8192 (round 6:
// the three plans of a Queen-class BiCG solve -- A, A^T, P A P^T -- spent 270 ms here).
template <int BLOCK, int PASS, int LOCAL_SORT, int DCAP = LOCAL_SORT / 2 + 256>
__global__ __launch_bounds__(BLOCK)
void csr_local_build(const v2i32 *__restrict__ blk, const int *__restrict__ idx, int cap, int ndmax,
                     int *__restrict__ nd_out, const int *__restrict__ doff, int *__restrict__ dcol,
                     unsigned short *__restrict__ lcol, int shift = 0)      // shift (PASS 0 only): count distinct (column >> shift) -- 4: the 128 B lines of x a block touches
{
    __shared__ int keys[LOCAL_SORT];
    __shared__ int dist[DCAP];
    __shared__ int wsum[BLOCK / WAVE];
    const int b = blockIdx.x, t = threadIdx.x;
    const int k0 = blk[b].y, k1 = blk[b + 1].y, cnt = k1 - k0;
    if (cnt <= 0 || cnt > cap || cnt > DCAP || cnt > LOCAL_SORT) { if (PASS == 0 && t == 0) nd_out[b] = 0; return; }
    if (PASS == 1 && doff[b + 1] == doff[b]) return;
    for (int i = t; i < LOCAL_SORT; i += BLOCK) keys[i] = i < cnt ? (idx[k0 + i] >> shift) : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= LOCAL_SORT; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < LOCAL_SORT; i += BLOCK) {
                const int l = i ^ j;
                if (l > i) {
                    const int a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    // distinct heads among the first cnt sorted keys: each thread owns LOCAL_SORT / BLOCK consecutive ones
    constexpr int PER = LOCAL_SORT / BLOCK;
    int mine = 0;
    for (int u = 0; u < PER; u++) {
        const int i = t * PER + u;
        if (i < cnt && (i == 0 || keys[i] != keys[i - 1])) mine++;
    }
    int incl = mine;                                 // inclusive scan over the workgroup: wave scan, then the wave totals
    const int lane = t & (WAVE - 1), w = t / WAVE;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const int v = __shfl_up(incl, off, WAVE); if (lane >= off) incl += v; }
    if (lane == WAVE - 1) wsum[w] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int q = 0; q < BLOCK / WAVE; q++) { if (q < w) base += wsum[q]; total += wsum[q]; }
    if (PASS == 0) { if (t == 0) nd_out[b] = total <= ndmax ? total : 0; return; }
    int pos = base + incl - mine;
    for (int u = 0; u < PER; u++) {
        const int i = t * PER + u;
        if (i < cnt && (i == 0 || keys[i] != keys[i - 1])) dist[pos++] = keys[i];
    }
    __syncthreads();
    const int nd = total, d0 = doff[b], padded = doff[b + 1] - d0;
    for (int j = t; j < padded; j += BLOCK) dcol[d0 + j] = dist[j < nd ? j : nd - 1];
    for (int i = t; i < cnt; i += BLOCK) {
        const int c = idx[k0 + i];
        int lo = 0, hi = nd - 1;                     // the column is in the list
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (dist[mid] < c) lo = mid + 1; else hi = mid; }
        lcol[k0 + i] = (unsigned short)lo;
    }
}

// ------------------------------------------------------------------------------ row-gather kernel, coded indices
// A matrix whose entries sit on at most 255 distinct diagonals (every structured-grid discretisation: the 7-point
// stencil has 7) does not need 4 B per column index: the plan stores ONE byte per non-zero, the position of
// (column - row) in a sorted dictionary of the offsets that occur (csr_collect_offsets / csr_encode below), and this
// kernel streams 9 B per non-zero instead of 12.  Same rows, same terms, same order as the kernel above (the column
// is rebuilt as row + dict[code]), so the sums are bit-identical; the index array itself stays in HBM for the
// paths that want it (long rows, array tails, transposition).
template <int BLOCK, int WORK, int U, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_coded_kernel(const int *__restrict__ ptr, const int *__restrict__ idx,
                           const double *__restrict__ val, const unsigned char *__restrict__ codes,
                           const int *__restrict__ dict, const double *__restrict__ x,
                           double *__restrict__ y, const v2i32 *__restrict__ blk,
                           int bfirst, int nb, Rows RW, int nnz_total,
                           const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                           const double *__restrict__ guard = nullptr, int pstride = 0, int xs_plane = 0)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    __shared__ double dot_scratch[BLOCK / WAVE];
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    constexpr int CAP = WORK + SLACK + 16;          // + the 16-entry alignment of the code slice
    __shared__ __attribute__((aligned(16))) double valL[(GUARD + CAP + 8 + 16) + 2 * WAVE];    // also the padded product stage of block_by_products
    __shared__ __attribute__((aligned(16))) unsigned char codeL[CAP + 16 + 16 * WAVE];
    __shared__ int dictL[256];

    const int lb = xcd_strip_unit((int)blockIdx.x, nb, xs_plane);      // (XCD strips: see xcd_strip_unit)
    Blk B = load_blk(blk, bfirst + lb);
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    const int ka = B.k0 & ~15;                      // 16 B aligned start of the code slice (128 B for the values)
    const int cnt = B.k1 - ka;
    const int np = (cnt + 1) >> 1;                  // 16 B pieces of the value slice
    const int nc = (cnt + 15) >> 4;                 // 16 B pieces of the code slice (the code array is padded)
    if (cnt > CAP || ka + 2 * np > nnz_total) {     // long row / last value of the array
        block_by_products<BLOCK, CAP, 4, DOT>(valL, ptr, idx, val, x, y, B, dots, acc0);
        __syncthreads();
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    for (int i = threadIdx.x; i < 256; i += BLOCK) dictL[i] = dict[i];
    const int rmine = B.r0 + (int)threadIdx.x;
    int s_first = 0, e_first = 0;
    if (rmine < B.r1) { s_first = ptr[rmine]; e_first = ptr[rmine + 1]; }
    {
        const int wbase = (int)threadIdx.x & ~(WAVE - 1), lane = (int)threadIdx.x & (WAVE - 1);
        for (int p0 = wbase; p0 < np; p0 += BLOCK) {
            const int p = min(p0 + lane, np - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
        }
        for (int q0 = wbase; q0 < nc; q0 += BLOCK) {
            const int q = min(q0 + lane, nc - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v4i32 *>(codes + ka) + q),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v4i32 *>(codeL) + q0), 16, 0, 2);
        }
    }
    __syncthreads();

    for (int r = rmine; r < B.r1; r += BLOCK) {
        int s = s_first, e = e_first;
        if (r != rmine) { s = ptr[r]; e = ptr[r + 1]; }
        const int len = e - s;
        const int off = s - ka;
        const double wr = dots.fetch(r);
        double acc = acc0;
        for (int j0 = 0; j0 < len; j0 += U) {
            int cc[U]; double vv[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = min(j0 + u, len - 1);         // clamped: repeats the row's last entry
                cc[u] = r + dictL[codeL[off + j]];
                vv[u] = valL[off + j];
            }
#pragma unroll
            for (int u = 0; u < U; u++) xx[u] = x[cc[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double t = vv[u] * xx[u];
                acc += (j0 + u < len) ? t : -0.0;           // -0.0 terms leave any sum bit-unchanged
            }
        }
        store_stream(y + r, acc);
        dots.add_loaded(wr, acc);
    }
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// ------------------------------------------------------------------------------ row-gather kernel, row patterns
// A structured-grid matrix repeats not only its offsets but whole ROWS of them: every interior row of the 7-point stencil is
// the same sequence of 7 offsets, the boundary rows are 26 more sequences.  When the rows of a coded matrix follow at most 255
// such patterns (liship_csr_plan_encode_row_patterns) the plan keeps ONE byte per ROW -- the pattern, which gives the row's
// length and its offsets from a small table in LDS -- instead of one byte per non-zero and a 4 B row pointer; the rows' starts
// are the running sum of the pattern lengths within the row block (a 2 B start per row is kept for blocks of more than BLOCK
// rows): the 7-point stencil streams 73 B per row (56 values, 1 pattern, 8 y, 8 x) instead of 83.  Same rows, same terms, same order as the two kernels above: bit-identical.
constexpr int PAT_TABLE = 1024 + 256 + 2;           // ints of LDS for the table: npat + 1 prefix entries, then the offsets
template <int BLOCK, int WORK, int U, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_pattern_kernel(const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
                             const unsigned char *__restrict__ rowpat, const unsigned short *__restrict__ rowrel,
                             const int *__restrict__ ptab, int tablen, int npat1,
                             const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                             int bfirst, int nb, Rows RW, int nnz_total,
                             const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                             const double *__restrict__ guard = nullptr, int pstride = 0)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    __shared__ double dot_scratch[BLOCK / WAVE];
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    constexpr int CAP = WORK + SLACK + 2;
    __shared__ __attribute__((aligned(16))) double valL[(GUARD + CAP + 8 + 16) + 2 * WAVE];    // also the product stage of block_by_products
    __shared__ int ptabL[PAT_TABLE];

    const int lb = blockIdx.x;
    Blk B = load_blk(blk, bfirst + lb);
    const int kplan = B.k0, rplan0 = B.r0, rplan1 = B.r1;   // the PLAN's block, whatever this launch clips
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    const int ka = B.k0 & ~1;                       // 16 B aligned start of the value slice
    const int cnt = B.k1 - ka;
    const int np = (cnt + 1) >> 1;                  // 16 B pieces of the value slice
    if (cnt > CAP || ka + 2 * np > nnz_total) {     // long row / last value of the array
        block_by_products<BLOCK, CAP, 4, DOT>(valL, ptr, idx, val, x, y, B, dots, acc0);
        __syncthreads();
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    for (int i = threadIdx.x; i < tablen; i += BLOCK) ptabL[i] = ptab[i];
    const int rmine = B.r0 + (int)threadIdx.x;
    int p_first = 0, s_first = 0;
    // Row starts without reading them: when the plan's block has at most BLOCK rows, lane t takes the pattern of the block's t-th
    // row, and the rows' starts are the running sum of the pattern lengths (wavefront scan + the wavefront totals through
    // LDS).  Larger blocks (very short rows) and launches clipped at the front read the 2 B starts.
    const bool scan = (rplan1 - rplan0) <= BLOCK && B.r0 == rplan0;      // (a launch clipped at its front reads the starts too)
    __shared__ int wtot[BLOCK / WAVE];
    if (scan) { if (rplan0 + (int)threadIdx.x < rplan1) p_first = rowpat[rplan0 + (int)threadIdx.x]; }
    else if (rmine < B.r1) { p_first = rowpat[rmine]; s_first = kplan + rowrel[rmine]; }
    {
        const int wbase = (int)threadIdx.x & ~(WAVE - 1), lane = (int)threadIdx.x & (WAVE - 1);
        for (int p0 = wbase; p0 < np; p0 += BLOCK) {
            const int p = min(p0 + lane, np - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
        }
    }
    __syncthreads();
    if (scan) {                                       // uniform
        const int lane = (int)threadIdx.x & (WAVE - 1), w = (int)threadIdx.x / WAVE;
        const int mylen = (rplan0 + (int)threadIdx.x < rplan1) ? ptabL[p_first + 1] - ptabL[p_first] : 0;
        int incl = mylen;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) { const int v = __shfl_up(incl, o, WAVE); if (lane >= o) incl += v; }
        if (lane == WAVE - 1) wtot[w] = incl;
        __syncthreads();
        int base = kplan;
        for (int q = 0; q < w; q++) base += wtot[q];
        s_first = base + incl - mylen;
    }

    for (int r = rmine; r < B.r1; r += BLOCK) {
        int pat = p_first, s = s_first;
        if (r != rmine) { pat = rowpat[r]; s = kplan + rowrel[r]; }
        const int ps = ptabL[pat], len = ptabL[pat + 1] - ps;
        const int *po = ptabL + npat1 + ps;
        const int off = s - ka;
        // <w,y> with w = x (CG's <p,Ap>, BiCGSTAB's <t,s>): w[r] is the x the row gathers for its diagonal entry anyway -- one
        // vector-memory instruction less per row; a row without a diagonal entry loads it after all
        const bool w_is_x = DOT >= 1 && wdot == x;
        double wr = w_is_x ? 0.0 : dots.fetch(r);
        bool have_w = !w_is_x;
        double acc = acc0;
        for (int j0 = 0; j0 < len; j0 += U) {
            int cc[U]; double vv[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = min(j0 + u, len - 1);         // clamped: repeats the row's last entry
                cc[u] = r + po[j];
                vv[u] = valL[off + j];
            }
#pragma unroll
            for (int u = 0; u < U; u++) xx[u] = x[cc[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double t = vv[u] * xx[u];
                acc += (j0 + u < len) ? t : -0.0;           // -0.0 terms leave any sum bit-unchanged
                if (DOT >= 1 && w_is_x && cc[u] == r) { wr = xx[u]; have_w = true; }
            }
        }
        if (DOT >= 1 && !have_w) wr = wdot[r];
        store_stream(y + r, acc);
        dots.add_loaded(wr, acc);
    }
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// ------------------------------------------------------------------------------ patterned rows of 8..32 entries: four lanes per row
// The 27-point stencil with varying coefficients (any matrix on <= 255 row patterns whose longest has 8..32 offsets, values
// streamed).  One lane per row leaves the general pattern kernel with 73 busy lanes of 256 and FOUR dependent gather round trips per
// row behind the barrier that waits for the value slice (27 gathers, 8 in flight): 0.47 of the roofline on its bytes.  Here a row
// is the business of a TEAM of four lanes: lane t of the team owns entries 8t .. 8t+7, so a row's gathers are one batch per lane and
// leave BEFORE the slice has landed (pattern byte -> 144 B pattern record -> gathers -> the slice's LDS-DMA behind them: the
// vector-memory counter counts in order, a wait for the records must not be a wait for the slice), every lane of the workgroup has work, and a workgroup lives for two
// long round trips instead of six.  The sum stays ONE chain per row, strictly left to right: lane 0 adds its eight products to the
// start value, hands the sum to lane 1 through the LDS crossbar (ds_bpermute: registers to registers, no memory -- the LDS round trip of
// the PRODUCTS is what sank round 2's team variant), and so on; a lane whose segment is empty adds -0.0 terms.  Same terms, same order: the
// reference's bits (lis_matvec_csr.c:97-109).  A wavefront takes 16 consecutive rows (no merge-path split: rows of 8..32 balance).
constexpr int TEAM_SEG = 8, TEAM_MAXLEN = 32, TEAM_REC = 9;     // a pattern record: 32 byte offsets (the tail repeats the last), length + 3 pad = 9 x 16 B
template <int BLOCK>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_pattern_team_kernel(const int *__restrict__ ptr, const double *__restrict__ val, const unsigned char *__restrict__ rowpat,
                                  const v4i32 *__restrict__ prec, const double *__restrict__ x, double *__restrict__ y, Rows RW, int nnz_total,
                                  const double *__restrict__ guard = nullptr)
{
    if (guard != nullptr && guard[0] != 0.0) return;               // (fused forms) device-driven Krylov loop already converged
    // A wavefront owns 16 consecutive rows AND their value slice: nothing is shared between the wavefronts of a workgroup, so there is no
    // barrier -- a wavefront waits for its own loads only.  lane = 16 t + i: the 16 lanes that hold segment t of 16 NEIGHBOURING rows
    // are neighbours, so a gather instruction touches four 128 B runs of x (with lane = 4 i + t no two neighbouring lanes shared a line:
    // 78 L1 accesses per gather instruction, the texture addresser 92 % busy -- profiles/r03_pattern_team_kernel.txt)
    constexpr int RPW = 16, STAGE = RPW * TEAM_MAXLEN + 2 + 46;  // rows, doubles of LDS per wavefront (16 B multiple)
    __shared__ __attribute__((aligned(16))) double stage[(BLOCK / WAVE) * STAGE];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (WAVE - 1);
    double *valL = stage + w * STAGE;
    const int r0 = RW.rb + ((int)blockIdx.x * (BLOCK / WAVE) + w) * RPW, r1 = min(r0 + RPW, RW.re);
    if (r0 >= r1) return;
    const int t = lane >> 4;
    const int r = min(r0 + (lane & 15), r1 - 1);                // (lanes beyond the last row repeat it and store nothing)
    const bool live = r0 + (lane & 15) < r1;
    const int k0 = ptr[r0], k1 = ptr[r1];                       // (uniform: scalar loads) the slice's extent, asked for first ...
    const int pat = rowpat[r];                                  // ... with everything the gathers wait for
    const int s = ptr[r];
    __builtin_amdgcn_sched_barrier(0);                          // (the scheduler would issue the scalar loads behind the wait for the pattern byte)
    const int ka = k0 & ~1, cnt = k1 - ka;
    int np = (cnt + 1) >> 1;                                    // 16 B pieces of the value slice
    const bool odd_end = ka + 2 * np > nnz_total;               // the last piece would pass the end of the array: its one value by a plain load
    if (odd_end) np--;
    const v4i32 *rec = prec + pat * TEAM_REC;
    const v4i32 o0 = rec[2 * t], o1 = rec[2 * t + 1];           // this lane's eight byte offsets ...
    const int len = *reinterpret_cast<const int *>(rec + 8);    // ... and the row's length: issued AHEAD of the slice
    const char *xb = reinterpret_cast<const char *>(x + r);
    double xx[TEAM_SEG];
    xx[0] = *reinterpret_cast<const double *>(xb + o0.x); xx[1] = *reinterpret_cast<const double *>(xb + o0.y);
    xx[2] = *reinterpret_cast<const double *>(xb + o0.z); xx[3] = *reinterpret_cast<const double *>(xb + o0.w);
    xx[4] = *reinterpret_cast<const double *>(xb + o1.x); xx[5] = *reinterpret_cast<const double *>(xb + o1.y);
    xx[6] = *reinterpret_cast<const double *>(xb + o1.z); xx[7] = *reinterpret_cast<const double *>(xb + o1.w);
    __builtin_amdgcn_sched_barrier(0);                          // (gathers, then the slice: a wait for the records must not be a wait for the slice)
#pragma unroll
    for (int it = 0; it < (RPW * TEAM_MAXLEN / 2 + WAVE) / WAVE; it++) {            // (at most five 1 KB pieces: no loop, no wait in front of it)
        const int p0 = it * WAVE;
        if (p0 + lane < np)                             // (lanes past the slice stay out: their 16 B would land in the next wavefront's stage)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p0 + lane),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
    }
    if (odd_end && lane == 0) valL[cnt - 1] = val[k1 - 1];
    __builtin_amdgcn_s_waitcnt(0);                              // this wavefront's slice (and gathers) have landed: vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nt = min(max(len - TEAM_SEG * t, 0), TEAM_SEG);   // entries of this lane's segment
    const double *vp = valL + (s - ka) + TEAM_SEG * t;          // (reads up to 7 doubles past a short row: inside the stage)
    double pm[TEAM_SEG];
#pragma unroll
    for (int u = 0; u < TEAM_SEG; u++) { const double pr = vp[u] * xx[u]; pm[u] = u < nt ? pr : -0.0; }
    double c = RW.acc0;
    const int from = ((lane - 16) & (WAVE - 1)) * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double in = RW.acc0;
        if (k > 0) in = __hiloint2double(__builtin_amdgcn_ds_bpermute(from, __double2hiint(c)),    // lane 16 t + i takes the sum of lane 16 (t-1) + i
                                         __builtin_amdgcn_ds_bpermute(from, __double2loint(c)));   // (the LDS crossbar: no memory is touched)
#pragma unroll
        for (int u = 0; u < TEAM_SEG; u++) in += pm[u];
        c = t == k ? in : c;
    }
    if (live && t == 3) store_stream(y + r, c);
}

// The same with x STAGED: profiles/r03_pattern_team_kernel.txt -- the kernel above is bound by the number of vector-memory instructions it issues
// (the texture addresser takes a 64-lane instruction at a fixed rate whatever the lanes ask for: half the gathers, 0.39 -> 0.33 ms; the same
// gathers with one lane in sixteen active, no change).  When one pattern carries most rows and its sorted offsets are runs of m consecutive
// columns (build_team_runs: the box stencils), the x a wavefront's 16 neighbouring rows need from a run are 15 + m consecutive doubles, and all
// runs together a few hundred bytes: ceil(slots / 64) coalesced loads -- issued at once, they depend on nothing but the row numbers -- put them
// in LDS, and an entry of a row reads its slot from there (an LDS read per entry instead of a gather per entry: 3 vector-memory instructions
// instead of 8 for the 27-point stencil).  Rows on other patterns whose offsets lie inside the dominant one's runs (the boundary rows of a
// stencil) read the same slots by their own records; rows with a foreign pattern gather for themselves.  Speculative addresses are clamped
// to [0, largest column].  Values, products, the chain through ds_bpermute: as above -- same terms, same order, the reference's bits.
struct TeamRuns { int nruns, slots, maxcol, maxlen; int start[16], base[16]; };     // run a: columns r0 + start[a] + position, slots base[a] .. base[a + 1])
template <int BLOCK, int NLOAD>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_pattern_team_staged_kernel(const int *__restrict__ ptr, const double *__restrict__ val, const unsigned char *__restrict__ rowpat,
                                         const v4i32 *__restrict__ prec, const v4i32 *__restrict__ pslot, const double *__restrict__ x,
                                         double *__restrict__ y, Rows RW, int nnz_total, const TeamRuns TR, int vcap, int xcap,
                                         const double *__restrict__ guard = nullptr)
{
    if (guard != nullptr && guard[0] != 0.0) return;               // (fused forms) device-driven Krylov loop already converged
    constexpr int RPW = 16;                                      // NLOAD = ceil(slots / 128): the loads are unconditional, so that the counter's waits can be exact
    extern __shared__ __attribute__((aligned(16))) double team_dyn[];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (WAVE - 1);
    double *valL = team_dyn + w * (vcap + xcap), *xL = valL + vcap;
    const int r0 = RW.rb + ((int)blockIdx.x * (BLOCK / WAVE) + w) * RPW, r1 = min(r0 + RPW, RW.re);
    if (r0 >= r1) return;
    const int t = lane >> 4, i = lane & 15;
    const int r = min(r0 + i, r1 - 1);
    const bool live = r0 + i < r1;
    const int k0 = ptr[r0], k1 = ptr[r1];
    const int pat = rowpat[r];
    __builtin_amdgcn_sched_barrier(0);                            // (the scalar loads and the pattern byte leave first)
    // the staged x: slot sl = base[run] + position holds column r0 + start[run] + position.  A lane takes TWO neighbouring slots with one
    // 16 B load (every run's width is even: a pair never straddles two runs; the address is a double's: v2f64u); a pair pushed inside the array by the
    // clamp at either end hands each slot the half that holds its column (the other slot of such a pair belongs to no stored entry)
    v2f64 xs[NLOAD];
#pragma unroll
    for (int k = 0; k < NLOAD; k++) {
        const int sl = 2 * (k * WAVE + lane);
        int st = TR.start[0], bs = 0;                            // the run the slot pair lies in (bases of unused runs are beyond every slot)
#pragma unroll
        for (int a = 1; a < 16; a++) { const bool in = sl >= TR.base[a]; st = in ? TR.start[a] : st; bs = in ? TR.base[a] : bs; }
        const int c = r0 + st + (sl - bs), cc = min(max(c, 0), TR.maxcol - 1);
        const v2f64 v = *reinterpret_cast<const v2f64u *>(x + cc);
        xs[k].x = c > cc ? v.y : v.x;                            // c == maxcol: its column is the pair's upper half
        xs[k].y = c < cc ? v.x : v.y;                            // c == -1: column 0 is the pair's lower half
    }
    __builtin_amdgcn_sched_barrier(0);
    const int ka = k0 & ~1, cnt = k1 - ka;
    int np = (cnt + 1) >> 1;
    const bool odd_end = ka + 2 * np > nnz_total;
    if (odd_end) np--;
    // this lane's record: eight slots (bytes), the row's length, whether its pattern is foreign to the runs -- one 16 B load, issued AHEAD of the slice (the
    // vector-memory counter counts in order).  (Measured and dropped: pattern bytes and records by scalar loads, one round per distinct pattern in the
    // wavefront -- two vector-memory instructions fewer and 5 % slower, 0.351 against 0.332 ms: the scalar chain sits in front of the slice.)
    const v4i32 rs = pslot[pat * 4 + t];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < (RPW * TEAM_MAXLEN / 2 + WAVE) / WAVE; it++) {
        const int p0 = it * WAVE;
        if (p0 + lane < np)                             // (lanes past the slice stay out: their 16 B would land in the next wavefront's stage)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(reinterpret_cast<const v2f64 *>(val + ka) + p0 + lane),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
    }
    if (odd_end && lane == 0) valL[cnt - 1] = val[k1 - 1];
    __builtin_amdgcn_sched_barrier(0);                            // (nothing that needs the record may move in between the slice's loads)
    asm volatile("" :: "v"(rs.w));                                 // (the record's unused word stays allocated: reusing its register while the load is in flight would wait for ALL loads)
    const int len = live ? (rs.z & 255) : 0;
    double xx[TEAM_SEG];
    const bool foreign = ((rs.z >> 8) & 255) != 0;
    // the row's start: the slice's start + the lengths of the rows before it among the wavefront's 16 (a scan inside the rows of 16 lanes: no
    // load of ptr[r], 4 B per row less)
    int incl = len;
    { int v;
      v = __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false); incl += v;     // row_shr:1
      v = __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false); incl += v;     // row_shr:2
      v = __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false); incl += v;     // row_shr:4
      v = __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false); incl += v; }   // row_shr:8
    const int s = k0 + incl - len;
    if (__any(foreign)) {                                          // rare: rows whose offsets the runs do not hold gather for themselves
        if (foreign) {
            const v4i32 *rec = prec + pat * TEAM_REC;
            const v4i32 o0 = rec[2 * t], o1 = rec[2 * t + 1];
            const char *xb = reinterpret_cast<const char *>(x + r);
            xx[0] = *reinterpret_cast<const double *>(xb + o0.x); xx[1] = *reinterpret_cast<const double *>(xb + o0.y);
            xx[2] = *reinterpret_cast<const double *>(xb + o0.z); xx[3] = *reinterpret_cast<const double *>(xb + o0.w);
            xx[4] = *reinterpret_cast<const double *>(xb + o1.x); xx[5] = *reinterpret_cast<const double *>(xb + o1.y);
            xx[6] = *reinterpret_cast<const double *>(xb + o1.z); xx[7] = *reinterpret_cast<const double *>(xb + o1.w);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                                 // everything this wavefront asked for has landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < NLOAD; k++) { const int sl = 2 * (k * WAVE + lane); if (sl < TR.slots) *reinterpret_cast<v2f64 *>(xL + sl) = xs[k]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!foreign) {
        const double *xr = xL + i;
        xx[0] = xr[rs.x & 255]; xx[1] = xr[(rs.x >> 8) & 255]; xx[2] = xr[(rs.x >> 16) & 255]; xx[3] = xr[(unsigned)rs.x >> 24];
        xx[4] = xr[rs.y & 255]; xx[5] = xr[(rs.y >> 8) & 255]; xx[6] = xr[(rs.y >> 16) & 255]; xx[7] = xr[(unsigned)rs.y >> 24];
    }
    const int nt = min(max(len - TEAM_SEG * t, 0), TEAM_SEG);
    const double *vp = valL + (s - ka) + TEAM_SEG * t;
    double pm[TEAM_SEG];
#pragma unroll
    for (int u = 0; u < TEAM_SEG; u++) { const double pr = vp[u] * xx[u]; pm[u] = u < nt ? pr : -0.0; }
    double c = RW.acc0;
    const int from = ((lane - 16) & (WAVE - 1)) * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double in = RW.acc0;
        if (k > 0) in = __hiloint2double(__builtin_amdgcn_ds_bpermute(from, __double2hiint(c)),
                                         __builtin_amdgcn_ds_bpermute(from, __double2loint(c)));
#pragma unroll
        for (int u = 0; u < TEAM_SEG; u++) in += pm[u];
        c = t == k ? in : c;
    }
    if (live && t == 3) store_stream(y + r, c);
}

// inclusive sum over the 64 lanes by data-parallel primitives: 4 shifts inside the rows of 16, then lane 15 of each row into
// the next row, then lane 31 into the upper half -- 6 adds, no LDS traffic.  All lanes must be active.
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// The same for matrices whose patterns have at most 7 offsets and no empty row (the 7-point stencil): a pattern is then ONE 32 B
// record -- the 7 column offsets in bytes (the tail repeats the last one) and the length -- and nothing of the row phase depends
// on LDS contents: a lane knows its columns from (row, pattern) alone and issues its x gathers BEFORE the value slice has
// landed.  With the codes the columns came out of the streamed slice and the gathers could only start behind it.
// The vector-memory counter counts IN ORDER: a wait for any load younger than the slice is a wait for the whole slice.  Hence
// the order here: the pattern bytes are loaded first (older than the slice: the wait for them leaves the slice in flight), the
// records come by SCALAR loads (their own counter; one load per distinct pattern in the wavefront, nearly always one), then the
// gathers leave, and the one barrier that ends the scan of the row lengths is the first wait behind the slice.
// The row phase is instruction-bound once everything overlaps (ablations in profiles/r02_csr_kernel_experiments.txt: with every
// load hitting a cache and no value stream it alone takes 0.70 ms at 512^3), so it is kept short: 32-bit byte offsets on a
// scalar base (one add per gather), unpredicated gathers, a 6-instruction scan of the lengths, and a wavefront whose rows all
// have 7 entries adds its products without selects.  Same terms, same order: bit-identical to every other CSR kernel.
// (Persistent workgroups that prefetch the next block's extents and pattern bytes were tried on top of this and were 10-30 %
// slower at every grid size -- DESIGN.md 5.)
constexpr int PAT7_MAX = 64;       // patterns the plan accepts for this kernel (more: the general pattern kernel)
template <int BLOCK, int WORK, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_pattern7_kernel(const int *__restrict__ ptr, const double *__restrict__ val,
                              const unsigned char *__restrict__ rowpat, const unsigned short *__restrict__ rowrel,
                              const v4i32 *__restrict__ ptab8,
                              const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                              int bfirst, int nb, Rows RW, int nnz_total,
                              const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                              const double *__restrict__ guard = nullptr, int pstride = 0, int xs_plane = 0)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged
    const int row_begin = RW.rb, row_end = RW.re;
    const double acc0 = RW.acc0;
    __shared__ double dot_scratch[BLOCK / WAVE];
    __shared__ int wtot[BLOCK / WAVE];
    constexpr int CAP = WORK + SLACK + 2;
    __shared__ __attribute__((aligned(16))) double valL[(GUARD + CAP + 8 + 16) + 2 * WAVE];
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    const int tid = (int)threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const bool w_is_x = DOT >= 1 && wdot == x;

    const int lb = xcd_strip_unit((int)blockIdx.x, nb, xs_plane);      // (XCD strips: see xcd_strip_unit)
    Blk B = load_blk(blk, bfirst + lb);
    const int kplan = B.k0, rplan0 = B.r0, rplan1 = B.r1;   // the PLAN's block, whatever this launch clips
    if (!clip_rows(B, ptr, row_begin, row_end)) {           // empty block: still owes its (zero) partial
        publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
        return;
    }
    const bool scan = (rplan1 - rplan0) <= BLOCK && B.r0 == rplan0;      // row starts = running sum of the lengths (see the kernel above)
    const int rmine = B.r0 + tid;
    const bool mine = rmine < B.r1;
    // 1. the pattern bytes (and, for a front-clipped block or one of more than BLOCK rows, the stored row starts): the oldest
    //    loads, unconditional (clamped index) so that no select or branch needs their value before the records do
    const int prow = scan ? rplan0 + tid : rmine;
    const bool planrow = scan ? prow < rplan1 : mine;
    // (issued by hand: the compiler would wait for this byte with vmcnt(0), i.e. for the whole slice issued behind it; the
    //  counter retires in order, so "at most NIT younger operations outstanding" is all the byte needs -- step 3)
    int pat;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(pat) : "v"(rowpat + min(prow, (scan ? rplan1 : B.r1) - 1)) : "memory");
    int s_first = scan ? 0 : kplan + (int)rowrel[min(rmine, B.r1 - 1)];
    // 2. the value slice, by a FIXED number of LDS-DMA instructions (the compiler can then wait for the pattern bytes with the
    //    slice still in flight): rows of 1..7 entries put at most 7/8 of a block's WORK + SLACK + 8 items into values; lanes
    //    past the slice re-read its last piece into stage slots nobody reads.  Only the matrix's very last value needs care:
    //    the 16 B piece that holds it would end past the array.
    const int ka = B.k0 & ~1;                       // 16 B aligned start of the value slice
    const int cnt = B.k1 - ka;
    const int np = (cnt + 1) >> 1;                  // 16 B pieces of the value slice
    const int tailv = (ka + 2 * np > nnz_total) ? 1 : 0;
    constexpr int MAXNNZ = (WORK + SLACK + 8) * 7 / 8 + 2, NIT = ((MAXNNZ + 1) / 2 + BLOCK - 1) / BLOCK;
    {
        static_assert(2 * NIT * BLOCK <= (GUARD + CAP + 8 + 16) + 2 * WAVE, "value stage too small");
        const int npd = np - tailv, wbase = tid & ~(WAVE - 1);
        const v2f64 *src = reinterpret_cast<const v2f64 *>(npd > 0 ? val + ka : val);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int p0 = wbase + it * BLOCK;
            const int p = max(min(p0 + lane, npd - 1), 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + p),
                (__attribute__((address_space(3))) void *)(reinterpret_cast<v2f64 *>(valL) + p0), 16, 0, 2);
        }
    }
    // 3. the records: one pair of scalar loads per distinct pattern among the wavefront's lanes
    int o[7] = {0, 0, 0, 0, 0, 0, 0}, len = 0;
    {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pat) : "n"(NIT) : "memory");       // the pattern byte is in; the slice stays in flight
        unsigned long long todo = __builtin_amdgcn_ballot_w64(planrow);
        while (todo != 0) {                                               // uniform
            int pw = __builtin_amdgcn_readlane(pat, (int)__builtin_ctzll(todo));
            asm volatile("" : "+s"(pw));                                  // (opaque: keeps the record address scalar)
            v4i32 a = ptab8[2 * pw], b = ptab8[2 * pw + 1];               // uniform address: scalar loads
            asm volatile("" : "+s"(a.x), "+s"(a.y), "+s"(a.z), "+s"(a.w), "+s"(b.x), "+s"(b.y), "+s"(b.z), "+s"(b.w));       // (and stay so)
            const bool hit = planrow && pat == pw;
            if (hit) { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; len = b.w; }
            todo &= ~__builtin_amdgcn_ballot_w64(hit);
        }
    }
    // 4. the gathers: behind nothing but the record
    double xx[7], wr = 0.0;
    const unsigned rb8 = (unsigned)rmine * 8u;            // byte offsets in 32 bits (the plan checks n < 2^29): scalar base + one add per gather
    if (mine) {
        if (DOT >= 1 && !w_is_x) wr = wdot[rmine];
#pragma unroll
        for (int u = 0; u < 7; u++) xx[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)o[u]));
    }
    double tail_value = 0.0;
    if (tailv && tid == 0) tail_value = val[B.k1 - 1];
    if (scan) {                                       // uniform
        const int incl = wave_inclusive_scan(len);
        if (lane == WAVE - 1) wtot[wv] = incl;
        __syncthreads();                              // the wavefront totals, the value slice, the gathers
        int base = kplan;
#pragma unroll
        for (int q = 0; q < BLOCK / WAVE - 1; q++) base += q < wv ? wtot[q] : 0;
        s_first = base + incl - len;
    } else __syncthreads();
    if (tailv) {                                      // uniform; behind the slice (whose clamped lanes land on this slot too)
        if (tid == 0) valL[cnt - 1] = tail_value;
        __syncthreads();
    }
    const bool full = __builtin_amdgcn_ballot_w64(mine && len != 7) == 0;       // wavefront-uniform: no row here is shorter than 7
    if (mine) {
        const double *vp = valL + (s_first - ka);
        double acc = acc0;
        if (full) {
#pragma unroll
            for (int u = 0; u < 7; u++) acc += vp[u] * xx[u];
        } else {
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const double t = vp[min(u, len - 1)] * xx[u];
                acc += (u < len) ? t : -0.0;            // -0.0 terms leave any sum bit-unchanged
            }
        }
        if (DOT >= 1 && w_is_x) {                     // w is x: the diagonal's gather has it (else the load below)
            bool have_w = false;
#pragma unroll
            for (int u = 0; u < 7; u++) if (u < len && o[u] == 0) { wr = xx[u]; have_w = true; }
            if (!have_w) wr = wdot[rmine];
        }
        store_stream(reinterpret_cast<double *>(reinterpret_cast<char *>(y) + rb8), acc);
        dots.add_loaded(wr, acc);
    }
    for (int r = rmine + BLOCK; r < B.r1; r += BLOCK) {        // blocks of more than BLOCK rows (very short rows): the later rows
        const int pt = rowpat[r], s = kplan + rowrel[r];
        const v4i32 a = ptab8[2 * pt], b = ptab8[2 * pt + 1];
        const int oo[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z}, ln = b.w, off = s - ka;
        const double wvl = dots.fetch(r);
        double xv[7], acc = acc0;
#pragma unroll
        for (int u = 0; u < 7; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)r * 8u + (unsigned)oo[u]));
#pragma unroll
        for (int u = 0; u < 7; u++) { const double t = valL[off + min(u, ln - 1)] * xv[u]; acc += (u < ln) ? t : -0.0; }
        store_stream(y + r, acc);
        dots.add_loaded(wvl, acc);
    }
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// Value records (CSR-VI in the literature: a matrix that holds few distinct values is stored by reference to them; here by whole
// rows): when every row of a pattern also carries the same VALUES -- a constant-coefficient stencil: the 27 patterns of the 7-point
// Laplacian are 27 (offsets, values) rows -- the plan keeps the 7 values beside the 7 offsets (96 B per pattern, checked bit for
// bit against every row at plan time) and the product needs neither the value nor the index stream: ONE byte per row.
// What is left per row is the pattern byte, 7 gathers of x and the store of y -- no value stage, no block geometry beyond
// the row blocks that keep the fused dots' partial slots where the other kernels put them.  The records sit in LDS (6 KB at
// most, loaded per workgroup while the pattern bytes are in flight); a lane reads its offsets, issues its gathers, and reads
// the values only when it adds.  Same products (the record holds the row's values bit for bit), same order: bit-identical.
// Measured on the way (512^3, profiles/r02_csr_kernel_experiments.txt): one block per workgroup with the record fetched by
// scalar loads 0.89-0.93 ms, of which 0.54 ms remain when every gather hits a cache and nothing is stored -- a chain of
// dependent round trips plus the cost of starting half a million short-lived workgroups; two blocks per workgroup 0.85 ms,
// four the same; XCD-contiguous block orders +-1 %.  256^3 (x within the Infinity Cache): 0.078 ms.
template <int BLOCK, int K, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerec_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, int npat,
                              const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                              int bfirst, int nb, Rows RW,
                              const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                              const double *__restrict__ guard = nullptr, int pstride = 0)
{
    const double stop = (DOT != 0 && guard != nullptr) ? guard[0] : 0.0;      // device-driven Krylov loop already converged? (looked at
    const double acc0 = RW.acc0;                                              // behind the record fill: not a step of the chain)
    __shared__ double dot_scratch[2 * K * (BLOCK / WAVE)];
    __shared__ __attribute__((aligned(16))) v4i32 recL[6 * PAT7_MAX];     // 96 B per pattern: 7 byte offsets + length, 7 values
    const int tid = (int)threadIdx.x;
    for (int t = tid; t < 6 * npat; t += BLOCK) recL[t] = rec[t];
    // a workgroup takes K consecutive row blocks at once, K rows per lane: with nothing to stream the kernel is a chain of
    // dependent round trips (extents -> pattern byte -> x -> y) plus the cost of starting a wavefront, and rows in flight per
    // wavefront are what hides both.  Each block keeps its own partial slot for the fused dots.
    const int lb0 = blockIdx.x * K;
    int r[K], r1[K], pat[K];
#pragma unroll
    for (int h = 0; h < K; h++) {
        if (DOT == 0) {                                 // the plain product owes nobody a partial per row block: chunks of BLOCK rows by
            const long long c0 = (long long)RW.rb + (long long)(lb0 + h) * BLOCK;          // arithmetic, one dependent load fewer
            r[h] = (int)min(c0, (long long)RW.re) + tid; r1[h] = (int)min(c0 + BLOCK, (long long)RW.re);
        } else {
            const Blk B = lb0 + h < nb ? load_blk(blk, bfirst + lb0 + h) : Blk{0, 0, 0, 0};
            r[h] = max(B.r0, RW.rb) + tid; r1[h] = min(B.r1, RW.re);    // only the rows matter here
        }
    }
#pragma unroll
    for (int h = 0; h < K; h++) pat[h] = r[h] < r1[h] ? (int)rowpat[r[h]] : -1;
    __syncthreads();                                                   // the records are in LDS (and the pattern bytes in)
    if (DOT != 0 && stop != 0.0) return;                               // (uniform; nothing has been written)
    RowDots<DOT> dots[K];
    double xx[K][7], wr[K];
    int len[K];
#pragma unroll
    for (int h = 0; h < K; h++) {                                      // all gathers first: 7 K per lane in flight
        dots[h] = RowDots<DOT>{wdot, 0.0, 0.0};
        len[h] = 0; wr[h] = 0.0;
        if (pat[h] >= 0) {
            const v4i32 a = recL[6 * pat[h]], b = recL[6 * pat[h] + 1];
            const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
            len[h] = b.w;
            const unsigned rb8 = (unsigned)r[h] * 8u;   // 32-bit byte offsets on a scalar base (the plan checks n + max offset < 2^29)
            if (DOT >= 1) wr[h] = wdot[r[h]];           // (w = x in CG: the line the diagonal's gather fetches anyway)
#pragma unroll
            for (int u = 0; u < 7; u++) xx[h][u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)o[u]));
        }
    }
#pragma unroll
    for (int h = 0; h < K; h++) {                                      // then the sums, the values read from LDS only now
        const bool full = __builtin_amdgcn_ballot_w64(pat[h] >= 0 && len[h] != 7) == 0;     // wavefront-uniform: no row here is shorter than 7
        if (pat[h] >= 0) {
            const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pat[h] + 2);
            const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
            double acc = acc0;
            if (full) {
#pragma unroll
                for (int u = 0; u < 7; u++) acc += v[u] * xx[h][u];
            } else {
#pragma unroll
                for (int u = 0; u < 7; u++) { const double t = v[u] * xx[h][u]; acc += (u < len[h]) ? t : -0.0; }   // -0.0 terms leave any sum bit-unchanged
            }
            store_stream(reinterpret_cast<double *>(reinterpret_cast<char *>(y) + (unsigned)r[h] * 8u), acc);
            dots[h].add_loaded(wr[h], acc);
        }
    }
#pragma unroll
    for (int h = 0; h < K; h++) {                                      // blocks of more than BLOCK rows (very short rows): the later rows
        for (int rr = r[h] + BLOCK; rr < r1[h]; rr += BLOCK) {
            const int pt = rowpat[rr];
            const v4i32 a = recL[6 * pt], b = recL[6 * pt + 1];
            const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pt + 2);
            const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z}, ln = b.w;
            const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
            const double wv = dots[h].fetch(rr);
            double xv[7], acc = acc0;
#pragma unroll
            for (int u = 0; u < 7; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)rr * 8u + (unsigned)o[u]));
#pragma unroll
            for (int u = 0; u < 7; u++) { const double t = v[u] * xv[u]; acc += (u < ln) ? t : -0.0; }
            store_stream(y + rr, acc);
            dots[h].add_loaded(wv, acc);
        }
    }
    if (DOT != 0) {                                   // the K blocks' partials behind ONE barrier; per block the order of publish_dots:
        const int lane = tid & (WAVE - 1), w = tid / WAVE, stride = pstride ? pstride : nb;      // wave butterfly, then the waves in order
#pragma unroll
        for (int h = 0; h < K; h++) {
            const double s0 = wave_sum(dots[h].c0), s1 = DOT >= 2 ? wave_sum(dots[h].c1) : 0.0;
            if (lane == 0) { dot_scratch[(2 * h) * (BLOCK / WAVE) + w] = s0; if (DOT >= 2) dot_scratch[(2 * h + 1) * (BLOCK / WAVE) + w] = s1; }
        }
        __syncthreads();
        if (tid < 2 * K && lb0 + tid / 2 < nb && (DOT >= 2 || (tid & 1) == 0)) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < BLOCK / WAVE; i++) t += dot_scratch[tid * (BLOCK / WAVE) + i];
            partial[(size_t)(tid & 1) * stride + lb0 + tid / 2] = t;
        }
    }
}

// The plain product with value records, two rows per lane.  The counters say what binds the kernel above (profiles/
// r02_valuerec_kernel_pmc.txt): 19 vector-memory instructions per wavefront, each of which the texture addresser takes apart
// in 16 quads of lanes whatever it hits -- 62 % of the kernel's cycles, TA busy 84 % -- while VALU is busy 20 % of the time and
// the x traffic is not felt at all.  An 8 B access per lane uses half of what a quad can carry.  So a lane takes the rows 2p and
// 2p + 1: when they share their pattern (all but two pairs per grid line) x[r + o] and x[r + 1 + o] are ONE 16 B load, y[r],
// y[r + 1] one 16 B store, the two pattern bytes one 2 B load: half the instructions per row.  A pair with two patterns (or a
// last row without a partner) takes the rows one after the other.  Same products, same order per row: bit-identical.
template <int BLOCK, int K>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerec_pair_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, int npat,
                                   const double *__restrict__ x, double *__restrict__ y, Rows RW)
{
    const double acc0 = RW.acc0;
    __shared__ __attribute__((aligned(16))) v4i32 recL[6 * PAT7_MAX];     // 96 B per pattern: 7 byte offsets + length, 7 values
    const int tid = (int)threadIdx.x;
    for (int t = tid; t < 6 * npat; t += BLOCK) recL[t] = rec[t];
    int ra[K], pa[K], pb[K];
#pragma unroll
    for (int h = 0; h < K; h++) {                         // pair (ra, ra + 1); pb < 0: no second row
        const long long c0 = (long long)RW.rb + ((long long)blockIdx.x * K + h) * (2 * BLOCK) + 2 * tid;
        ra[h] = (int)min(c0, (long long)RW.re);
        pa[h] = pb[h] = -1;
        if (ra[h] + 1 < RW.re) {
            if ((ra[h] & 1) == 0) { const unsigned two = *reinterpret_cast<const unsigned short *>(rowpat + ra[h]); pa[h] = (int)(two & 255u); pb[h] = (int)(two >> 8); }
            else { pa[h] = rowpat[ra[h]]; pb[h] = rowpat[ra[h] + 1]; }
        } else if (ra[h] < RW.re) pa[h] = rowpat[ra[h]];
    }
    __syncthreads();                                      // the records are in LDS
    v2f64 xx[K][7];
#pragma unroll
    for (int h = 0; h < K; h++) {                         // the paired gathers first: 7 K 16 B loads per lane in flight
        if (pa[h] >= 0 && pa[h] == pb[h]) {
            const v4i32 a = recL[6 * pa[h]], b = recL[6 * pa[h] + 1];
            const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
            const unsigned rb8 = (unsigned)ra[h] * 8u;
#pragma unroll
            for (int u = 0; u < 7; u++) xx[h][u] = *reinterpret_cast<const v2f64u *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)o[u]));
        }
    }
#pragma unroll
    for (int h = 0; h < K; h++) {
        if (pa[h] >= 0 && pa[h] == pb[h]) {
            const v4i32 b = recL[6 * pa[h] + 1];
            const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pa[h] + 2);
            const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
            const int len = b.w;
            double s0 = acc0, s1 = acc0;
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const double t0 = v[u] * xx[h][u].x, t1 = v[u] * xx[h][u].y;
                s0 += (u < len) ? t0 : -0.0;              // -0.0 terms leave any sum bit-unchanged
                s1 += (u < len) ? t1 : -0.0;
            }
            v2f64 out; out.x = s0; out.y = s1;
            if ((ra[h] & 1) == 0) store_stream(reinterpret_cast<v2f64 *>(y + ra[h]), out);
            else { store_stream(y + ra[h], s0); store_stream(y + ra[h] + 1, s1); }
        } else {
#pragma unroll
            for (int w = 0; w < 2; w++) {                 // two patterns in the pair, or a single last row: one row at a time
                const int pt = w ? pb[h] : pa[h], r = ra[h] + w;
                if (pt < 0) continue;
                const v4i32 a = recL[6 * pt], b = recL[6 * pt + 1];
                const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pt + 2);
                const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z}, len = b.w;
                const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
                double xv[7], acc = acc0;
#pragma unroll
                for (int u = 0; u < 7; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)r * 8u + (unsigned)o[u]));
#pragma unroll
                for (int u = 0; u < 7; u++) { const double t = v[u] * xv[u]; acc += (u < len) ? t : -0.0; }
                store_stream(y + r, acc);
            }
        }
    }
}

// Value records, the dominant pattern speculated (round 3).  The kernels above are a chain of dependent round trips per workgroup
// -- records -> LDS, pattern byte -> barrier -> record -> x -> y -- and the counters say that chain, not bytes, is what a product
// costs (profiles/r02_valuerec_kernel_pmc.txt: with six of seven gathers removed the time does not move).  But nearly every row of
// a structured matrix carries ONE pattern -- the interior row: 98.8 % of the rows at 512^3 -- and the other patterns of a stencil are
// SUBSEQUENCES of it (a boundary row is the interior row minus the neighbours that do not exist).  So the plan names the dominant
// pattern D (a histogram of the pattern bytes), hands its seven byte offsets and values to the kernel as ARGUMENTS (scalar
// registers: no load, no LDS, no barrier), and a lane issues the x gathers of D for its rows at once, together with the load of
// its pattern bytes: ONE round trip.  When the bytes arrive, a wavefront whose rows all carry D (most) multiplies by the scalar
// values and stores.  Otherwise the lanes with another pattern p take p's DOMINANT-SLOT record from a 64 B-per-pattern table --
// which of D's seven slots p has (mask) and p's values in those slots -- by scalar loads, one per distinct pattern among the
// wavefront's lanes (a waterfall over readlane; the table is 1.7 KB and lives in the scalar cache), and add the slots they have in
// D's order, which is their own order: same products, same order, bit-identical.  A pattern that is not a subsequence of D (mask
// bit 7: the ghost-column rows of a partitioned matrix) and the wavefronts whose speculative addresses would leave x (the first and
// last |max offset| rows) take their rows one by one with their own records, as the kernels above do.
typedef double v8f64 __attribute__((ext_vector_type(8)));
struct DomTile { int S, cshift, ntiled, nfull; };       // spmv_csr_valuerec_dom_kernel: tiled lane -> row mapping (S = 0: none)
struct DomRec { int off[7]; int pat; double val[7]; int mask, d0; };      // byte offsets and values of the dominant pattern, its pattern byte, its slots (length), the slot of offset 0 (-1: none)

// slots of D that pattern `pt` (uniform) has, and its values there: 8 doubles per pattern, [0] = the mask in the low word
__device__ __forceinline__ void dom_waterfall(const double *__restrict__ drec, const DomRec &D, int pt, bool valid, double (&v)[7], unsigned &m)
{
#pragma unroll
    for (int u = 0; u < 7; u++) v[u] = D.val[u];
    m = (unsigned)D.mask;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid && pt != D.pat);
    while (todo != 0) {                                   // (uniform) one scalar record load per distinct pattern among the lanes
        const int l = __builtin_ctzll(todo);
        const int q = __builtin_amdgcn_readlane(pt, l);
        const v8f64 R = *reinterpret_cast<const v8f64 *>(drec + 8 * q);      // (uniform address: one s_load_dwordx16)
        const bool me = valid && pt == q;
        m = me ? (unsigned)__double2loint(R[0]) : m;
#pragma unroll
        for (int u = 0; u < 7; u++) v[u] = me ? R[1 + u] : v[u];
        todo &= ~__builtin_amdgcn_ballot_w64(me);
    }
}

// A pattern may END in up to six entries (the row itself, +0.0) behind its slots of the dominant pattern: the padding of an ELL row laid out row by row
// (lis_matrix_ell.c:1035-1042 pads with value 0, index i; lis_matvec_ell.c:113-128 adds those terms last).  Bits 8..10 of the mask count them; each adds
// 0.0 * x[row] -- a signed zero, or a NaN from an infinite x: the term is formed, not assumed.  Without this the boundary rows of a constant-coefficient stencil
// in ELL storage were "foreign" and took their own records: 256^3 0.087 ms against 0.050 for the other formats.
__device__ __forceinline__ double dom_pad_terms(double s, unsigned m, const double *__restrict__ x, int row)
{
    const unsigned k = (m >> 8) & 7u;
    if (__builtin_amdgcn_ballot_w64(k != 0) != 0) {                  // (uniform) rare: x[row] is read again here rather than picked out of the gathers
        const double t = 0.0 * x[row];
#pragma unroll
        for (unsigned q = 0; q < 6; q++) s += (q < k) ? t : -0.0;
    }
    return s;
}

// sum across the 64 lanes by DPP moves alone (no LDS permute): the rows of 16 by the xor partners 8, 4, 2, 1, then lane 15 of a row broadcast into the next row
// (rows 1 and 3) and lane 31 into rows 2 and 3 -- the total is valid in the lanes of row 3 (48..63).  NOT wave_sum's order: for epilogues with sums of their own.
__device__ __forceinline__ double wave_sum_row3(double v)
{
    v += lane_xor_in_row<8>(v);
    v += lane_xor_in_row<4>(v);
    v += lane_xor_in_row<2>(v);
    v += lane_xor_in_row<1>(v);
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, false);
        v += __hiloint2double(hi, lo);
    }
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x143, 0xc, 0xf, false);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// Fused dots of a workgroup WITHOUT a barrier at its end (which would hold every wavefront's slot until the slowest has its lines back): a wavefront parks
// its lanes' sums in LDS and counts itself in; the wavefront that counts in last adds the NW wavefronts' values lane by lane, in wavefront order, folds its
// 64 lanes (one butterfly per workgroup instead of one per wavefront: the epilogue's VALU instructions were what it cost) and writes the partial.  The same
// bits whoever is last.  `count` must have been zeroed behind a barrier at the kernel's START, where the wavefronts arrive together.
template <int BLOCK, int DOT>
__device__ __forceinline__ void workgroup_dots_last(double c0, double c1, double *part, unsigned *count, double *__restrict__ partial, int slot, int stride, bool write)
{
    constexpr int NW = BLOCK / WAVE;
    const int tid = (int)threadIdx.x, wbase = tid & ~(WAVE - 1);
    part[tid] = c0;
    if (DOT >= 2) part[BLOCK + tid] = c1;
    unsigned seen = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // every lane's sums are in LDS before the wavefront counts itself in
    if (tid == wbase + WAVE - 1)
        seen = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    seen = (unsigned)__builtin_amdgcn_readlane((int)seen, WAVE - 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // ... and the last one reads the others' behind its count
    if (seen == NW - 1 && write) {                                    // (uniform)
        const int l = tid - wbase;
        double e = 0.0, f = 0.0;
#pragma unroll
        for (int i = 0; i < NW; i++) e += part[i * WAVE + l];
        if (DOT >= 2) {
#pragma unroll
            for (int i = 0; i < NW; i++) f += part[BLOCK + i * WAVE + l];
        }
        e = wave_sum_row3(e);
        if (DOT >= 2) f = wave_sum_row3(f);
        if (l == WAVE - 1) {
            partial[slot] = e;
            if (DOT >= 2) partial[(size_t)stride + slot] = f;
        }
    }
}

// one row by its own record (96 B, global): the kernels above, without LDS.  Rare (ghost-column rows, the matrix's first and last rows), so it is written for
// few registers, not for latency: slots 0..3, then slots 4..6 -- inlined in kernels that hold seven 16 B gathers, 38 live registers here cost them their occupancy
__device__ __forceinline__ double own_record_row(const v4i32 *__restrict__ rec, int pt, int r, const double *__restrict__ x, double acc0)
{
    const v4i32 b = rec[6 * pt + 1];
    const int len = b.w;
    const v2f64 *q = reinterpret_cast<const v2f64 *>(rec + 6 * pt + 2);
    double acc = acc0;
    {
        const v4i32 a = rec[6 * pt];
        const v2f64 q0 = q[0], q1 = q[1];
        const int o[4] = {a.x, a.y, a.z, a.w};
        const double v[4] = {q0.x, q0.y, q1.x, q1.y};
        double xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)r * 8u + (unsigned)o[u]));
#pragma unroll
        for (int u = 0; u < 4; u++) { const double t = v[u] * xv[u]; acc += (u < len) ? t : -0.0; }
    }
    if (__builtin_amdgcn_ballot_w64(len > 4) != 0) {                 // (uniform)
        const v2f64 q2 = q[2], q3 = q[3];
        const int o[3] = {b.x, b.y, b.z};
        const double v[3] = {q2.x, q2.y, q3.x};
        double xv[3];
#pragma unroll
        for (int u = 0; u < 3; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)r * 8u + (unsigned)o[u]));
#pragma unroll
        for (int u = 0; u < 3; u++) { const double t = v[u] * xv[u]; acc += (4 + u < len) ? t : -0.0; }
    }
    return acc;
}

template <int BLOCK, int DOT = 0>
__global__ __launch_bounds__(BLOCK, 8)                  // 8 wavefronts per SIMD (64 VGPRs) with the dots' registers too: the product lives on the misses it keeps in flight
void spmv_csr_valuerec_dom_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, const double *__restrict__ drec,
                                  const DomRec D, int safe_lo, int safe_hi,
                                  const double *__restrict__ x, double *__restrict__ y, Rows RW, int run, const DomTile TL,
                                  const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                                  const double *__restrict__ guard = nullptr, int pstride = 0, int wslot = -1, int total = 0)
{
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;   // device-driven Krylov loop already converged (nothing has been written)
    const double acc0 = RW.acc0;
    const int tid = (int)threadIdx.x;
    const int wbase = tid & ~(WAVE - 1);
    double c0 = 0.0, c1 = 0.0;
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    if (DOT != 0) {                                                   // the only barrier: at the start, where the wavefronts of a workgroup arrive together
        if (tid == 0) dot_count = 0u;
        __syncthreads();
    }
    // workgroup -> chunk of rows.  Workgroups go round-robin over the 8 XCDs; with run > 1 each XCD walks runs of `run` consecutive
    // chunks, so that the x lines neighbouring chunks share (the +-n neighbours of a grid line) are fetched by ONE L2
    int chunk = (int)blockIdx.x;
    if (run > 1) { const int xcd = chunk % NUM_XCD, slot = chunk / NUM_XCD; chunk = ((slot / run) * NUM_XCD + xcd) * run + slot % run; }
    {
        int rw, ra;                                                   // the wavefront's lowest row, the lane's first row
        bool safe;                                                    // (uniform) speculation is safe: every address r*8 + offset of the wavefront's rows (16 B loads: one more) lies inside x[0, n)
        bool live = true;                                             // (uniform) the wavefront has rows
        if (TL.S > 0 && chunk < TL.ntiled) {
            // TILED rows: the workgroup's 256 lane pairs cover T = 256 >> cshift lines of 2 << cshift columns each, the lines S rows apart
            // (S: the stride of the pattern's middle offsets, +-n of a 3-D stencil) -- a wavefront then gathers its rows' +-S neighbours from
            // lines its OWN diagonal gathers fetch (L1 hits instead of L2 requests), and the tile's halo is two lines for T instead of two per line
            const int wpg = TL.S >> (TL.cshift + 1);                   // workgroups per group of T lines
            const int g = chunk / wpg, cb = chunk - g * wpg;
            const int T = (BLOCK >> TL.cshift);
            const int base = RW.rb + g * T * TL.S + (cb << (TL.cshift + 1));
            const int t = tid >> TL.cshift, cp = tid & ((1 << TL.cshift) - 1);
            ra = base + t * TL.S + 2 * cp;
            const int t0 = wbase >> TL.cshift, t1 = (wbase + WAVE - 1) >> TL.cshift;
            rw = base + t0 * TL.S;
            safe = rw >= safe_lo && base + t1 * TL.S + (2 << TL.cshift) + 2 <= safe_hi;
        } else {
            const long long c0w = (long long)RW.rb + (TL.S > 0 ? TL.nfull : 0) + (long long)(chunk - (TL.S > 0 ? TL.ntiled : 0)) * (2 * BLOCK) + 2 * wbase;
            if (c0w >= RW.re) live = false;                           // (uniform)
            rw = live ? (int)c0w : RW.rb;
            ra = rw + 2 * (tid - wbase);
            safe = live && rw >= safe_lo && rw + 2 * WAVE <= safe_hi && rw + 2 * WAVE <= RW.re;
        }
        if (safe) {
            unsigned two;
            if ((RW.rb & 1) == 0) two = *reinterpret_cast<const unsigned short *>(rowpat + ra);
            else two = (unsigned)rowpat[ra] | ((unsigned)rowpat[ra + 1] << 8);
            const unsigned rb8 = (unsigned)ra * 8u;
            v2f64 xx[7], ww;
            ww.x = ww.y = 0.0;
            if (DOT != 0 && wslot < 0) {                              // (uniform) w is not x, or the dominant pattern has no diagonal entry
                if ((RW.rb & 1) == 0) ww = *reinterpret_cast<const v2f64 *>(wdot + ra);
                else { ww.x = wdot[ra]; ww.y = wdot[ra + 1]; }
            }
#pragma unroll
            for (int u = 0; u < 7; u++) xx[u] = *reinterpret_cast<const v2f64u *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)D.off[u]));
            if (DOT != 0) {                                           // (uniform jump: seven selects cost 28 instructions a wavefront)
                switch (wslot) {
                case 0: ww = xx[0]; break;
                case 1: ww = xx[1]; break;
                case 2: ww = xx[2]; break;
                case 3: ww = xx[3]; break;
                case 4: ww = xx[4]; break;
                case 5: ww = xx[5]; break;
                case 6: ww = xx[6]; break;
                default: break;
                }
            }
            const int pa = (int)(two & 255u), pb = (int)(two >> 8);
            double s0 = acc0, s1 = acc0;
            if (__builtin_amdgcn_ballot_w64(pa != D.pat || pb != D.pat) == 0) {      // (uniform) every row here is the dominant pattern
                if (D.mask == 0x7f) {
#pragma unroll
                    for (int u = 0; u < 7; u++) { s0 += D.val[u] * xx[u].x; s1 += D.val[u] * xx[u].y; }
                } else {
#pragma unroll
                    for (int u = 0; u < 7; u++) {
                        const double t0 = D.val[u] * xx[u].x, t1 = D.val[u] * xx[u].y;
                        s0 += ((D.mask >> u) & 1) ? t0 : -0.0;
                        s1 += ((D.mask >> u) & 1) ? t1 : -0.0;
                    }
                }
            } else {
                double v[7];
                unsigned m;
                dom_waterfall(drec, D, pa, true, v, m);
                if (m & 0x80u) s0 = own_record_row(rec, pa, ra, x, acc0);
                else {
#pragma unroll
                    for (int u = 0; u < 7; u++) { const double t = v[u] * xx[u].x; s0 += ((m >> u) & 1u) ? t : -0.0; }    // -0.0 terms leave any sum bit-unchanged
                    s0 = dom_pad_terms(s0, m, x, ra);
                }
                dom_waterfall(drec, D, pb, true, v, m);
                if (m & 0x80u) s1 = own_record_row(rec, pb, ra + 1, x, acc0);
                else {
#pragma unroll
                    for (int u = 0; u < 7; u++) { const double t = v[u] * xx[u].y; s1 += ((m >> u) & 1u) ? t : -0.0; }
                    s1 = dom_pad_terms(s1, m, x, ra + 1);
                }
            }
            v2f64 out; out.x = s0; out.y = s1;
            if ((RW.rb & 1) == 0) store_stream(reinterpret_cast<v2f64 *>(y + ra), out);
            else { store_stream(y + ra, s0); store_stream(y + ra + 1, s1); }
            if (DOT >= 1) { c0 += ww.x * s0; c0 += ww.y * s1; }
            if (DOT >= 2) { c1 += s0 * s0; c1 += s1 * s1; }
        } else if (live) {                                            // the matrix's first and last rows, a launch's tail: row by row
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const int r = ra + w;
                if (r < RW.re) {
                    const double acc = own_record_row(rec, (int)rowpat[r], r, x, acc0);
                    store_stream(y + r, acc);
                    if (DOT >= 1) c0 += wdot[r] * acc;
                    if (DOT >= 2) c1 += acc * acc;
                }
            }
        }
    }
    if (DOT != 0)         // one partial per workgroup (tile), no barrier here.  (A partial per wavefront -- a million 8 B stores at 512^3 -- cost 0.05 ms; persistent
        workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, chunk, pstride ? pstride : total, chunk < total);      // workgroups spill: 0.9 ms)
}

// Z-MARCHING form of the dominant-pattern product (round 4) for the 7-point stencil on a grid whose lines are a multiple of 128 long: offsets exactly
// {-SO, -S, -1, 0, +1, +S, +SO} in ascending slot order (S: a grid line, SO: a plane).  The kernel above reads every x seven times through L1 (seven 16 B loads per lane
// pair; counters: TA busy 97 %, ~89 requests in flight per CU -- latency / issue bound at 0.58 of the roofline).  Here a workgroup owns a tile of 128 columns x TY
// lines of a plane and WALKS `zseg` planes: a plane's tile (+ one halo line above and below, one halo column left and right) is loaded ONCE by coalesced 16 B loads
// issued D planes ahead and parked in registers, written to one of two LDS buffers, and the row sums take -1 / 0 / +1 / -S / +S from LDS and -SO / +SO from the
// registers of the plane before and the plane after: 1.3 loads of 16 B per lane pair and plane instead of seven, one barrier per plane.  512^3: 0.49 -> 0.41 ms
// (0.58 -> 0.70 of 8 TB/s on the 17 B per row).  Same terms in the same order: rows on other patterns take their masks and values by the waterfall above (-0.0 terms),
// foreign rows their own records, ELL's padding terms as above -- y is the reference's, bit for bit.  Speculative addresses (the planes before the first and after the
// last, the halo of the grid's faces) are clamped into x[0, nx); their values only ever meet masked slots.
struct DomMarch { int S, SO, tiles_x, tiles_y, zseg, nseg, z0, z1, wgs, xcd, perm, order, planes, modes; };
struct BoxAlt { double v[7]; };       // (BOX, ALT) by slot: the value of a slot whose neighbour lies outside the grid when the rows keep it with a value of its own (DIA's explicit zeros)      // planes [z0, z1) of the grid (whole planes of the launch's row range); perm: 3 bits per slot, which neighbour it is (0: -SO, 1: -S, 2: -1, 3: 0, 4: +1, 5: +S, 6: +SO)
// masks of the rows' patterns over the dominant one's slots, by scalar loads, one round per distinct pattern among the lanes that `need` one (the first double of a
// pattern's record: spmv_csr_valuerec_march_kernel's short form, whose plan has no other kind of pattern)
__device__ __forceinline__ unsigned dom_masks_only(const double *__restrict__ drec, const DomRec &D, int pt, bool need, unsigned m)
{
    unsigned long long todo = __builtin_amdgcn_ballot_w64(need);
    while (todo != 0) {                                   // (uniform)
        const int q = __builtin_amdgcn_readlane(pt, __builtin_ctzll(todo));
        const unsigned mq = q == D.pat ? (unsigned)D.mask : (unsigned)__double2loint(drec[8 * q]);      // (uniform address: one scalar load)
        const bool me = need && pt == q;
        m = me ? mq : m;
        todo &= ~__builtin_amdgcn_ballot_w64(me);
    }
    return m;
}

// ORD: the slot order -- 0 ascending columns, 1 the reference's generators' (-SO, +SO, -S, +S, -1, +1, 0: lis's test drivers), 2 any (M.perm).  WS: w is a vector of its
// own (else the diagonal's pair).  GEN: the plan has patterns that are more than a mask (values of their own, foreign patterns, ELL's padding terms): the waterfall of
// the gathering kernel, every plane; else (the Poisson matrix: the faces' patterns are masks) a lane keeps its rows' masks from plane to plane and asks only when a
// pattern byte changes (the first and last planes).
// BOX: plan time has checked, row by row, that in the planes of this launch a slot is missing exactly where its neighbour lies outside the grid (dom_box_check) and
// that the faces' rows carry the dominant pattern's values: then no pattern byte is read at all, and instead of masks the halo cells that only masked slots ever read
// (the column left of the grid's first, the line above its first line, the plane before its first plane ... and their opposites) hold a ZERO whose sign makes the product
// with the slot's value -0.0, the term every masked slot adds: the sums run the unmasked code everywhere.  16 B per row: x once, y once.
// What the rows of a box do with a neighbour outside the grid is one of three things per side (M.modes, plan time): the slot is missing (the signed zero above), it is there
// with the dominant value and a real x behind it (a multi-rank job's ghost plane: nothing to do), or it is there with another value (ALT: DIA keeps the diagonals' explicit
// zeros, 0.0 * x of the row across the line's end -- the value is picked per row, x is the real one).  PADS: a (row, +0.0) term behind the sum per missing slot (ELL's padding).
template <int LPW, int D_, int DOT, bool WS, int ORD, bool GEN, bool BOX = false, bool ALT = false, bool PADS = false, bool PT = true>
__global__ __launch_bounds__(256)
void spmv_csr_valuerec_march_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, const double *__restrict__ drec,
                                    const DomRec D, const double *__restrict__ x, double *__restrict__ y, double acc0, const DomMarch M, int nx,
                                    const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                                    const double *__restrict__ guard = nullptr, int pstride = 0, const BoxAlt A = BoxAlt{})
{
    constexpr int BLOCK = 256, TX = 128, TY = 4 * LPW, LX = TX + 4;      // an LDS line: [pad][left halo][TX columns][right halo][pad]: a lane's pair at 2 + 2 lane, 16 B aligned
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;      // device-driven Krylov loop already converged (nothing has been written)
    __shared__ __attribute__((aligned(16))) double buf[2][(TY + 2) * LX];
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    const int tid = (int)threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & (WAVE - 1);
    if (DOT != 0 && tid == 0) dot_count = 0u;                         // (a barrier follows before anyone counts itself in)
    int wg = (int)blockIdx.x;
    const int ntile = M.tiles_x * M.tiles_y;
    if (M.xcd) { const int e = M.wgs / NUM_XCD; if (wg < e * NUM_XCD) { const int k = wg % NUM_XCD, j = wg / NUM_XCD; wg = k * e + j; } }      // each XCD a contiguous eighth of the (segment, tile) list: its L2 holds the halo lines neighbouring tiles share (the last wgs % 8 keep their place)
    const int seg = wg / ntile, t = wg - seg * ntile;
    const int ty = t / M.tiles_x, tx = t - ty * M.tiles_x;
    const int za = M.z0 + seg * M.zseg, zb = min(M.z1, za + M.zseg);
    const int col0 = tx * TX, line0 = ty * TY;
    const long long S = M.S, SO = M.SO;
    const int S_ = M.S;
    // lines that are not a multiple of 128 long (round 5): the last tile of a line holds SX < 128 columns (even, >= 4: dom_march_shape).  Its lanes beyond the last pair
    // repeat that pair's loads (no traffic of their own), park nothing in LDS, store nothing and add nothing to the dots; the right halo sits behind the last pair.
    // PT = false (grids of whole tiles, the plain box form: the headline's instantiation) folds every one of these tests away -- they cost 128^3 12 %, 512^3 1.3 %.
    const int SX = PT ? min(TX, (int)S_ - col0) : TX, hl = PT ? (SX >> 1) - 1 : WAVE - 1, lc = PT ? 2 * min(lane, hl) : 2 * lane;
    const bool act = PT ? lane <= hl : true;
    // ... and planes whose lines are not a multiple of TY: the last tile of a plane holds LY < TY lines.  A wavefront's lines beyond them repeat the last line's loads,
    // park nothing, store nothing; the bottom halo line sits behind the last line (LDS row LY + 1), loaded and parked by wavefront 3 as ever.
    const int LY = PT ? min(TY, (int)(SO / S) - line0) : TY;
    auto lval = [&](int i) { return PT ? w * LPW + i < LY : true; };
    auto roff = [&](int i) { return (long long)(line0 + (PT ? min(w * LPW + i, LY - 1) : w * LPW + i)) * S + col0 + lc; };      // this lane's pair of line i inside a plane
    auto at = [&](long long a) { return (int)(a < 0 ? 0 : (a > (long long)nx - 2 ? (long long)nx - 2 : a)); };      // (pairs: the last start is nx - 2)
    // (BOX) the zero that a masked slot's x is replaced by: its product with the slot's value must be -0.0, so it carries the opposite of the value's sign
    auto poison = [&](int kind) {
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 7; u++) if ((int)((M.perm >> (3 * u)) & 7) == kind) v = D.val[u];
        return __builtin_signbit(v) ? 0.0 : -0.0;
    };
    const int lines = (int)(SO / S), planes_all = M.planes;
    const bool box_left = BOX && tx == 0, box_right = BOX && tx == M.tiles_x - 1, box_top = BOX && ty == 0, box_bottom = BOX && ty == M.tiles_y - 1;      // (uniform)
    auto mode = [&](int kind) { return (M.modes >> (2 * kind)) & 3; };      // 0: the slot is missing, 1: there with the dominant value, 2: there with A.v's
    (void)lines;
    struct Packet { v2f64 own[LPW]; v2f64 hy; double hx[LPW]; v2f64 ww[WS ? LPW : 1]; unsigned short pat[LPW]; };
    auto load_packet = [&](Packet &P, int z, bool pats) {
        const long long pz = (long long)z * SO;
#pragma unroll
        for (int i = 0; i < LPW; i++) P.own[i] = *reinterpret_cast<const v2f64u *>(x + at(pz + roff(i)));
        if (w == 0) P.hy = *reinterpret_cast<const v2f64u *>(x + at((long long)z * SO + (long long)(line0 - 1) * S + col0 + lc));
        if (w == 3) P.hy = *reinterpret_cast<const v2f64u *>(x + at((long long)z * SO + (long long)(line0 + LY) * S + col0 + lc));
        if (lane == 0 || lane == hl) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.hx[i] = x[at((long long)z * SO + (long long)(line0 + (PT ? min(w * LPW + i, LY - 1) : w * LPW + i)) * S + (lane == 0 ? col0 - 1 : col0 + SX))];
        }
        if (pats && z < M.z1) {                                       // (uniform) the pattern bytes (and w, when it is a vector of its own) of the rows this plane's sums are for
            if (!BOX) {
#pragma unroll
                for (int i = 0; i < LPW; i++) P.pat[i] = *reinterpret_cast<const unsigned short *>(rowpat + pz + roff(i));
            }
            if (DOT != 0 && WS) {
#pragma unroll
                for (int i = 0; i < LPW; i++) P.ww[i] = *reinterpret_cast<const v2f64u *>(wdot + pz + roff(i));
            }
        }
    };
    auto store_packet = [&](const Packet &P, double *B) {
        if (act) {
#pragma unroll
            for (int i = 0; i < LPW; i++) if (lval(i)) *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + 2 + 2 * lane) = P.own[i];
            if (w == 0) { v2f64 h = P.hy; if (box_top && mode(1) == 0) { h.x = h.y = poison(1); } *reinterpret_cast<v2f64 *>(B + 2 + 2 * lane) = h; }
            if (w == 3) { v2f64 h = P.hy; if (box_bottom && mode(5) == 0) { h.x = h.y = poison(5); } *reinterpret_cast<v2f64 *>(B + (LY + 1) * LX + 2 + 2 * lane) = h; }
        }
        if (lane == 0 || lane == hl) {
#pragma unroll
            for (int i = 0; i < LPW; i++) if (lval(i)) {
                double h = P.hx[i];
                if (box_left && lane == 0 && mode(2) == 0) h = poison(2);
                if (box_right && lane == hl && mode(4) == 0) h = poison(4);
                B[(w * LPW + i + 1) * LX + (lane == 0 ? 1 : 2 + SX)] = h;
            }
        }
    };
    Packet Q[D_];
    v2f64 prev[LPW];
    unsigned short pat0[LPW];
    v2f64 ww0[WS ? LPW : 1];
    unsigned patc[LPW], mc[LPW];                                      // (short form) the pattern bytes a lane last asked about and its rows' masks, first row | second row << 8
#pragma unroll
    for (int i = 0; i < LPW; i++) { patc[i] = (unsigned)D.pat | ((unsigned)D.pat << 8); mc[i] = (unsigned)D.mask | ((unsigned)D.mask << 8); }
    {   // prologue: plane za - 1 (registers only), plane za (LDS + its pattern bytes), planes za + 1 .. za + D in flight
        Packet P;
        load_packet(P, za - 1, false);
#pragma unroll
        for (int i = 0; i < LPW; i++) { prev[i] = P.own[i]; if (BOX && za == 0 && mode(0) == 0) { prev[i].x = prev[i].y = poison(0); } }      // (uniform) the grid's first plane has no plane before it
        load_packet(P, za, true);
        store_packet(P, buf[za & 1]);
#pragma unroll
        for (int i = 0; i < LPW; i++) pat0[i] = BOX ? (unsigned short)0 : P.pat[i];
        if (WS) {
#pragma unroll
            for (int i = 0; i < LPW; i++) ww0[i] = P.ww[i];
        }
    }
#pragma unroll
    for (int d = 0; d < D_; d++) load_packet(Q[d], za + 1 + d, true);
    __syncthreads();
    double c0 = 0.0, c1 = 0.0;
    for (int zq = za; zq < zb; zq += D_) {
#pragma unroll
        for (int d = 0; d < D_; d++) {
            const int z = zq + d;
            if (z < zb) {                                             // (uniform)
                Packet &P = Q[d];                                     // plane z + 1: the oldest in flight
                store_packet(P, buf[(z + 1) & 1]);
                const double *B = buf[z & 1];
#pragma unroll
                for (int i = 0; i < LPW; i++) {
                    const int li = (w * LPW + i + 1) * LX + 2 + 2 * lane;
                    const v2f64 c = *reinterpret_cast<const v2f64 *>(B + li);
                    const double l = B[li - 1], r = B[li + 2];
                    const v2f64 up = *reinterpret_cast<const v2f64 *>(B + li - LX), dn = *reinterpret_cast<const v2f64 *>(B + li + LX);
                    v2f64 xl, xr, xx[7], nxt = P.own[i];
                    xl.x = l; xl.y = c.x; xr.x = c.y; xr.y = r;
                    if (BOX && z == planes_all - 1 && mode(6) == 0) { nxt.x = nxt.y = poison(6); }      // (uniform) the grid's last plane has no plane behind it
                    if (ORD == 0) { xx[0] = prev[i]; xx[1] = up; xx[2] = xl; xx[3] = c; xx[4] = xr; xx[5] = dn; xx[6] = nxt; }
                    else if (ORD == 1) { xx[0] = prev[i]; xx[1] = nxt; xx[2] = up; xx[3] = dn; xx[4] = xl; xx[5] = xr; xx[6] = c; }
                    else {
#pragma unroll
                        for (int u = 0; u < 7; u++)
                            switch ((M.perm >> (3 * u)) & 7) {      // (uniform)
                            case 0: xx[u] = prev[i]; break;
                            case 1: xx[u] = up; break;
                            case 2: xx[u] = xl; break;
                            case 3: xx[u] = c; break;
                            case 4: xx[u] = xr; break;
                            case 5: xx[u] = dn; break;
                            default: xx[u] = nxt; break;
                            }
                    }
                    const long long row = (long long)z * SO + roff(i);
                    const bool live = act && lval(i);
                    const int ra = (int)row;
                    const int pa = (int)(pat0[i] & 255u), pb = (int)(pat0[i] >> 8);
                    double s0 = acc0, s1 = acc0;
                    if (BOX) {
                        // which sides of the grid this pair's rows lie on (x: the first row of lane 0 / the second row of the last lane; y, z: uniform)
                        const bool o_l = box_left && lane == 0, o_r = box_right && lane == hl;
                        const bool o_u = box_top && w * LPW + i == 0, o_d = box_bottom && w * LPW + i == LY - 1, o_p = z == 0, o_n = z == planes_all - 1;
#pragma unroll
                        for (int u = 0; u < 7; u++) {
                            double v0 = D.val[u], v1 = D.val[u];
                            if (ALT) {
                                const int k = ORD == 0 ? u : (u == 0 ? 0 : u == 1 ? 6 : u == 2 ? 1 : u == 3 ? 5 : u == 4 ? 2 : u == 5 ? 4 : 3);      // the slot's neighbour (compile time)
                                if (mode(k) == 2) {                             // (uniform)
                                    const bool out0 = k == 0 ? o_p : k == 1 ? o_u : k == 2 ? o_l : k == 4 ? false : k == 5 ? o_d : k == 6 ? o_n : false;
                                    const bool out1 = k == 0 ? o_p : k == 1 ? o_u : k == 2 ? false : k == 4 ? o_r : k == 5 ? o_d : k == 6 ? o_n : false;
                                    v0 = out0 ? A.v[u] : v0; v1 = out1 ? A.v[u] : v1;
                                }
                            }
                            s0 += v0 * xx[u].x; s1 += v1 * xx[u].y;
                        }
                        if (PADS) {                                             // a term 0.0 * x[row] per missing slot, behind the sum (lis_matvec_ell.c:113-128 adds the padding last)
                            const int ku = (o_u && mode(1) == 0) + (o_d && mode(5) == 0) + (o_p && mode(0) == 0) + (o_n && mode(6) == 0);
                            const int k0 = ku + (o_l && mode(2) == 0), k1 = ku + (o_r && mode(4) == 0);
                            if (__builtin_amdgcn_ballot_w64((k0 | k1) != 0) != 0) {      // (uniform)
                                const double t0 = 0.0 * c.x, t1 = 0.0 * c.y;
#pragma unroll
                                for (int q = 0; q < 3; q++) { s0 += (q < k0) ? t0 : -0.0; s1 += (q < k1) ? t1 : -0.0; }      // (a row lies on at most three sides)
                            }
                        }
                    } else if (!GEN) {
                        if (__builtin_amdgcn_ballot_w64((unsigned)pat0[i] != patc[i]) != 0) {      // (uniform) a pattern byte changed since the last plane: ask again
                            unsigned ma = dom_masks_only(drec, D, pa, pa != (int)(patc[i] & 255u), mc[i] & 255u);
                            unsigned mb = dom_masks_only(drec, D, pb, pb != (int)(patc[i] >> 8), mc[i] >> 8);
                            mc[i] = (ma & 255u) | ((mb & 255u) << 8);
                            patc[i] = pat0[i];
                        }
                        const unsigned full = (unsigned)D.mask | ((unsigned)D.mask << 8);
                        if (__builtin_amdgcn_ballot_w64(mc[i] != full) == 0) {      // (uniform) every row here has all of the dominant pattern's slots
#pragma unroll
                            for (int u = 0; u < 7; u++) { s0 += D.val[u] * xx[u].x; s1 += D.val[u] * xx[u].y; }
                        } else {
                            const unsigned m = mc[i];
#pragma unroll
                            for (int u = 0; u < 7; u++) {
                                const double t0 = D.val[u] * xx[u].x, t1 = D.val[u] * xx[u].y;
                                s0 += ((m >> u) & 1u) ? t0 : -0.0;              // -0.0 terms leave any sum bit-unchanged
                                s1 += ((m >> (8 + u)) & 1u) ? t1 : -0.0;
                            }
                        }
                    } else if (__builtin_amdgcn_ballot_w64(pa != D.pat || pb != D.pat) == 0) {      // (uniform) every row here is the dominant pattern
#pragma unroll
                        for (int u = 0; u < 7; u++) { s0 += D.val[u] * xx[u].x; s1 += D.val[u] * xx[u].y; }
                    } else {
                        double v[7];
                        unsigned m;
                        dom_waterfall(drec, D, pa, true, v, m);
                        if (m & 0x80u) s0 = own_record_row(rec, pa, ra, x, acc0);
                        else {
#pragma unroll
                            for (int u = 0; u < 7; u++) { const double tt = v[u] * xx[u].x; s0 += ((m >> u) & 1u) ? tt : -0.0; }
                            s0 = dom_pad_terms(s0, m, x, ra);
                        }
                        dom_waterfall(drec, D, pb, true, v, m);
                        if (m & 0x80u) s1 = own_record_row(rec, pb, ra + 1, x, acc0);
                        else {
#pragma unroll
                            for (int u = 0; u < 7; u++) { const double tt = v[u] * xx[u].y; s1 += ((m >> u) & 1u) ? tt : -0.0; }
                            s1 = dom_pad_terms(s1, m, x, ra + 1);
                        }
                    }
                    v2f64 out; out.x = s0; out.y = s1;
                    if (live) store_stream(reinterpret_cast<v2f64 *>(y + row), out);
                    if (DOT != 0 && live) {
                        const v2f64 wv = WS ? ww0[i] : c;
                        c0 += wv.x * s0; c0 += wv.y * s1;
                        if (DOT >= 2) { c1 += s0 * s0; c1 += s1 * s1; }
                    }
                    prev[i] = c;
                    if (!BOX) pat0[i] = P.pat[i];
                    if (WS) ww0[i] = P.ww[i];
                }
                load_packet(Q[d], z + 1 + D_, true);                  // this slot's next plane
                __syncthreads();                                      // plane z + 1 is in LDS, plane z's buffer is free
            }
        }
    }
    if (DOT != 0) workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, (int)blockIdx.x, pstride ? pstride : M.wgs, true);
}

// The fused dots, FOUR rows per lane: one wavefront per row block, no LDS, no barrier.  Lane q of the wavefront holds the virtual
// lanes 4q .. 4q + 3 of the block's 256 (rows base + 4q + j) in four accumulators.  The one-row kernels' tree on them: the butterfly
// steps 32, 16, 8, 4 pair virtual lanes of equal j whose physical lanes are 8, 4, 2, 1 apart -- all inside a row of 16 lanes (DPP
// moves, no LDS permute) --, the steps 2 and 1 add inside the lane: (e0 + e2) + (e1 + e3); a row of 16 lanes is a virtual wavefront,
// and lane 0 adds the four rows' totals in order.  Every addition has the operands it has in the one-row kernels, so the partials are
// the same bits (tests/golden/reduction_bits.json, the forms of test_spmv_csr_index_codes).
template <int BLOCK, int DOT>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerec_dom_dot4_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, const double *__restrict__ drec,
                                       const DomRec D, int safe_lo, int safe_hi,
                                       const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                                       int bfirst, int nb, Rows RW,
                                       const double *__restrict__ wdot, double *__restrict__ partial,
                                       const double *__restrict__ guard, int pstride, int wslot)
{
    const double stop = guard != nullptr ? guard[0] : 0.0;          // device-driven Krylov loop already converged? (a scalar load beside the extents')
    const double acc0 = RW.acc0;
    const int tid = (int)threadIdx.x, q = tid & (WAVE - 1);
    // one row block per wavefront (stacking the workgroup's blocks a grid line apart, as the plain product's tiles do, was measured: -2 %)
    const int lb = (int)blockIdx.x * (BLOCK / WAVE) + __builtin_amdgcn_readfirstlane(tid / WAVE);
    if (lb >= nb) return;                                 // (uniform per wavefront; no barrier anywhere below)
    const Blk B = load_blk(blk, bfirst + lb);
    const int r0 = max(B.r0, RW.rb), r1 = min(B.r1, RW.re);
    if (stop != 0.0) return;
    double c0[4] = {0.0, 0.0, 0.0, 0.0}, c1[4] = {0.0, 0.0, 0.0, 0.0};
    for (int base = r0; base < r1; base += 256) {         // (uniform) virtual lane t holds the rows base + t
        const int ra = base + 4 * q;
        const bool four = ra + 3 < r1;
        const bool safe = base >= safe_lo && base + 4 * WAVE <= safe_hi;          // (uniform) every speculative address stays inside x[0, n)
        bool done = false;
        if (safe) {
            unsigned pats = 0;
            v2f64 xx[2][7], ww[2];
            if (four) {
                if ((ra & 3) == 0) pats = *reinterpret_cast<const unsigned *>(rowpat + ra);
                else pats = (unsigned)rowpat[ra] | ((unsigned)rowpat[ra + 1] << 8) | ((unsigned)rowpat[ra + 2] << 16) | ((unsigned)rowpat[ra + 3] << 24);
                if (wslot < 0) {                          // (uniform) w is not x, or the dominant pattern has no diagonal entry
                    if ((ra & 1) == 0) { ww[0] = *reinterpret_cast<const v2f64 *>(wdot + ra); ww[1] = *reinterpret_cast<const v2f64 *>(wdot + ra + 2); }
                    else { ww[0].x = wdot[ra]; ww[0].y = wdot[ra + 1]; ww[1].x = wdot[ra + 2]; ww[1].y = wdot[ra + 3]; }
                }
                const unsigned rb8 = (unsigned)ra * 8u;
#pragma unroll
                for (int u = 0; u < 7; u++) {
                    xx[0][u] = *reinterpret_cast<const v2f64u *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)D.off[u]));
                    xx[1][u] = *reinterpret_cast<const v2f64u *>(reinterpret_cast<const char *>(x) + (rb8 + 16u + (unsigned)D.off[u]));
                }
#pragma unroll
                for (int u = 0; u < 7; u++) if (u == wslot) { ww[0] = xx[0][u]; ww[1] = xx[1][u]; }      // w = x (CG's <p, A p>): the diagonal's gather IS w
            }
            double sv[4] = {acc0, acc0, acc0, acc0};
            if (__builtin_amdgcn_ballot_w64(four && pats != (unsigned)D.pat * 0x01010101u) == 0) {      // (uniform) every row here is the dominant pattern
                if (four) {
#pragma unroll
                    for (int u = 0; u < 7; u++) {
                        const bool on = (D.mask >> u) & 1;            // (uniform; all seven for a full-length pattern)
                        const double t0 = D.val[u] * xx[0][u].x, t1 = D.val[u] * xx[0][u].y, t2 = D.val[u] * xx[1][u].x, t3 = D.val[u] * xx[1][u].y;
                        sv[0] += on ? t0 : -0.0; sv[1] += on ? t1 : -0.0; sv[2] += on ? t2 : -0.0; sv[3] += on ? t3 : -0.0;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    double v[7];
                    unsigned m;
                    const int pj = (int)((pats >> (8 * j)) & 255u);
                    dom_waterfall(drec, D, pj, four, v, m);
                    if (four) {
                        if (m & 0x80u) sv[j] = own_record_row(rec, pj, ra + j, x, acc0);
                        else {
#pragma unroll
                            for (int u = 0; u < 7; u++) {
                                const double xv = (j & 1) ? xx[j >> 1][u].y : xx[j >> 1][u].x;
                                const double t = v[u] * xv;
                                sv[j] += ((m >> u) & 1u) ? t : -0.0;          // -0.0 terms leave any sum bit-unchanged
                            }
                            sv[j] = dom_pad_terms(sv[j], m, x, ra + j);
                        }
                    }
                }
            }
            if (four) {
                v2f64 o0, o1; o0.x = sv[0]; o0.y = sv[1]; o1.x = sv[2]; o1.y = sv[3];
                if ((ra & 1) == 0) { store_stream(reinterpret_cast<v2f64 *>(y + ra), o0); store_stream(reinterpret_cast<v2f64 *>(y + ra + 2), o1); }
                else { store_stream(y + ra, sv[0]); store_stream(y + ra + 1, sv[1]); store_stream(y + ra + 2, sv[2]); store_stream(y + ra + 3, sv[3]); }
                c0[0] += ww[0].x * sv[0]; c0[1] += ww[0].y * sv[1]; c0[2] += ww[1].x * sv[2]; c0[3] += ww[1].y * sv[3];
                if (DOT >= 2) { c1[0] += sv[0] * sv[0]; c1[1] += sv[1] * sv[1]; c1[2] += sv[2] * sv[2]; c1[3] += sv[3] * sv[3]; }
                done = true;
            }
        }
        if (!done) {                                      // a block's last rows, the matrix's first and last rows: row by row, their own records
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = ra + j;
                if (r < r1) {
                    const double acc = own_record_row(rec, (int)rowpat[r], r, x, acc0);
                    store_stream(y + r, acc);
                    c0[j] += wdot[r] * acc;
                    if (DOT >= 2) c1[j] += acc * acc;
                }
            }
        }
    }
    const int stride = pstride ? pstride : nb;
#pragma unroll
    for (int res = 0; res < (DOT >= 2 ? 2 : 1); res++) {
        double e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double t = res ? c1[j] : c0[j];
            t += lane_xor_in_row<8>(t);
            t += lane_xor_in_row<4>(t);
            t += lane_xor_in_row<2>(t);
            t += lane_xor_in_row<1>(t);
            e[j] = t;
        }
        const double vw = (e[0] + e[2]) + (e[1] + e[3]);          // a virtual wavefront's sum, in every lane of its row of 16
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(vw), 16 * i), hi = __builtin_amdgcn_readlane(__double2hiint(vw), 16 * i);
            t += __hiloint2double(hi, lo);
        }
        if (q == 0) partial[(size_t)res * stride + lb] = t;
    }
}

// The fused-dot forms two rows per lane.  The partial of a row block is DEFINED by the one-row kernels: lane t of the block's
// 256 holds sum_k w*y over its rows r0 + t + 256 k, a wavefront adds its 64 lanes by the butterfly 32, 16, 8, 4, 2, 1, lane 0
// adds the four wavefronts in order.  Here lane p of a block's 128 holds the "virtual lanes" 2p and 2p + 1 in two accumulators,
// and the same tree is walked on them: the butterfly steps 32 .. 2 pair virtual lanes of equal parity 16 .. 1 physical lanes
// apart (each accumulator by itself, inside a half wavefront of 32), the last step adds the lane's two accumulators (a + b is
// b + a), and a half wavefront is a virtual wavefront.  Every addition has the operands it has in the one-row kernels: the
// partials -- and with them every dot, every iteration count -- are bit-identical (tests/golden/reduction_bits.json and the
// A/B forms of test_spmv_csr_index_codes pin it).  A workgroup of 256 lanes takes two row blocks.
template <int BLOCK, int DOT>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerec_pair_dot_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, int npat,
                                       const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                                       int bfirst, int nb, Rows RW,
                                       const double *__restrict__ wdot, double *__restrict__ partial,
                                       const double *__restrict__ guard, int pstride)
{
    static_assert(BLOCK == 256, "two row blocks of 128 lane pairs");
    const double stop = guard != nullptr ? guard[0] : 0.0;          // device-driven Krylov loop already converged? (looked at behind the
    const double acc0 = RW.acc0;                                    // record fill: the flag's round trip is not a step of the chain)
    __shared__ double scratch[2][2][4];                   // [block of the workgroup][result][virtual wavefront]
    __shared__ __attribute__((aligned(16))) v4i32 recL[6 * PAT7_MAX];
    const int tid = (int)threadIdx.x;
    for (int t = tid; t < 6 * npat; t += BLOCK) recL[t] = rec[t];
    const int h = __builtin_amdgcn_readfirstlane(tid >> 7), p = tid & 127;        // wavefronts 0, 1: the first block; 2, 3: the second
    const int lb = blockIdx.x * 2 + h;
    const Blk B = lb < nb ? load_blk(blk, bfirst + lb) : Blk{0, 0, 0, 0};
    const int r0 = max(B.r0, RW.rb), r1 = min(B.r1, RW.re);
    __syncthreads();                                      // the records are in LDS
    if (stop != 0.0) return;                              // (uniform; nothing has been written)
    double c0[2] = {0.0, 0.0}, c1[2] = {0.0, 0.0};
    for (int base = r0; base < r1; base += 256) {         // (uniform per wavefront) virtual lane t holds the rows base + t
        const int ra = base + 2 * p;
        int pa = -1, pb = -1;
        if (ra + 1 < r1) {
            if ((ra & 1) == 0) { const unsigned two = *reinterpret_cast<const unsigned short *>(rowpat + ra); pa = (int)(two & 255u); pb = (int)(two >> 8); }
            else { pa = rowpat[ra]; pb = rowpat[ra + 1]; }
        } else if (ra < r1) pa = rowpat[ra];
        if (pa >= 0 && pa == pb) {
            const v4i32 a = recL[6 * pa], b = recL[6 * pa + 1];
            const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z}, len = b.w;
            const unsigned rb8 = (unsigned)ra * 8u;
            v2f64 xx[7], ww;
            if ((ra & 1) == 0) ww = *reinterpret_cast<const v2f64 *>(wdot + ra); else { ww.x = wdot[ra]; ww.y = wdot[ra + 1]; }
#pragma unroll
            for (int u = 0; u < 7; u++) xx[u] = *reinterpret_cast<const v2f64u *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)o[u]));
            const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pa + 2);
            const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
            double s0 = acc0, s1 = acc0;
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const double t0 = v[u] * xx[u].x, t1 = v[u] * xx[u].y;
                s0 += (u < len) ? t0 : -0.0;              // -0.0 terms leave any sum bit-unchanged
                s1 += (u < len) ? t1 : -0.0;
            }
            v2f64 out; out.x = s0; out.y = s1;
            if ((ra & 1) == 0) store_stream(reinterpret_cast<v2f64 *>(y + ra), out);
            else { store_stream(y + ra, s0); store_stream(y + ra + 1, s1); }
            c0[0] += ww.x * s0; c0[1] += ww.y * s1;
            if (DOT >= 2) { c1[0] += s0 * s0; c1[1] += s1 * s1; }
        } else {
#pragma unroll
            for (int w = 0; w < 2; w++) {                 // two patterns in the pair, or a single last row: one row at a time
                const int pt = w ? pb : pa, r = ra + w;
                if (pt < 0) continue;
                const v4i32 a = recL[6 * pt], b = recL[6 * pt + 1];
                const v2f64 *q = reinterpret_cast<const v2f64 *>(recL + 6 * pt + 2);
                const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const int o[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z}, len = b.w;
                const double v[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x};
                const double wr = wdot[r];
                double xv[7], acc = acc0;
#pragma unroll
                for (int u = 0; u < 7; u++) xv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned)r * 8u + (unsigned)o[u]));
#pragma unroll
                for (int u = 0; u < 7; u++) { const double t = v[u] * xv[u]; acc += (u < len) ? t : -0.0; }
                store_stream(y + r, acc);
                c0[w] += wr * acc;
                if (DOT >= 2) c1[w] += acc * acc;
            }
        }
    }
    // the one-row kernels' tree on the virtual lanes (see above)
    const int vw = p >> 5;                                // virtual wavefront of this lane's two virtual lanes
#pragma unroll
    for (int res = 0; res < (DOT >= 2 ? 2 : 1); res++) {
        double e = res ? c1[0] : c0[0], f = res ? c1[1] : c0[1];
        e += __shfl_xor(e, 16, WAVE); f += __shfl_xor(f, 16, WAVE);
        e += lane_xor_in_row<8>(e);   f += lane_xor_in_row<8>(f);
        e += lane_xor_in_row<4>(e);   f += lane_xor_in_row<4>(f);
        e += lane_xor_in_row<2>(e);   f += lane_xor_in_row<2>(f);
        e += lane_xor_in_row<1>(e);   f += lane_xor_in_row<1>(f);
        const double t = e + f;
        if ((p & 31) == 0) scratch[h][res][vw] = t;
    }
    __syncthreads();
    if (tid < 4) {                                        // thread = (block of the workgroup, result)
        const int hh = tid >> 1, res = tid & 1, slot = blockIdx.x * 2 + hh, stride = pstride ? pstride : nb;
        if (slot < nb && (res == 0 || DOT >= 2)) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) t += scratch[hh][res][i];
            partial[(size_t)res * stride + slot] = t;
        }
    }
}

// Value records for longer rows (up to 32 entries: the 9-point stencil in 2-D, the 19- and 27-point ones in 3-D, whose 27 row
// patterns carry their values when the coefficients are constant).  One row per lane; the records -- 144 B of byte offsets and
// the length, 256 B of values per pattern -- sit in LDS; a row is walked in chunks of 8 entries: offsets, 8 gathers in flight,
// values, 8 additions in order (terms beyond the row's length add -0.0).  Same products in the same order: bit-identical.
constexpr int PATW_MAX = 128, PATW_LEN = 32, PATW_OFF = 36;            // patterns (48 until round 4; the b x b blocking of a 7-point stencil has 27 b: 54, 81, 108), entries per pattern, ints per offset record
template <int BLOCK, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerecw_kernel(const unsigned char *__restrict__ rowpat, const v4i32 *__restrict__ rec, int npat,
                               const double *__restrict__ x, double *__restrict__ y, const v2i32 *__restrict__ blk,
                               int bfirst, int nb, Rows RW,
                               const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                               const double *__restrict__ guard = nullptr, int pstride = 0)
{
    const double stop = (DOT != 0 && guard != nullptr) ? guard[0] : 0.0;
    const double acc0 = RW.acc0;
    __shared__ double dot_scratch[BLOCK / WAVE];
    extern __shared__ __attribute__((aligned(16))) double recw_dyn[];     // npat x 256 B of values, then npat x 144 B of offsets (the launcher sizes it: 400 B per pattern)
    double *valL = recw_dyn;
    int *offL = reinterpret_cast<int *>(recw_dyn + (size_t)npat * PATW_LEN);
    const int tid = (int)threadIdx.x;
    {   // the image: npat x 144 B of offsets, then npat x 256 B of values
        const v4i32 *src = rec;
        for (int t = tid; t < npat * (PATW_OFF / 4); t += BLOCK) reinterpret_cast<v4i32 *>(offL)[t] = src[t];
        src += npat * (PATW_OFF / 4);
        for (int t = tid; t < npat * (PATW_LEN / 2); t += BLOCK) reinterpret_cast<v4i32 *>(valL)[t] = src[t];
    }
    RowDots<DOT> dots{wdot, 0.0, 0.0};
    const int lb = blockIdx.x;
    const Blk B = load_blk(blk, bfirst + lb);
    const int r0 = max(B.r0, RW.rb), r1 = min(B.r1, RW.re);
    __syncthreads();                                      // the records are in LDS
    if (DOT != 0 && stop != 0.0) return;                  // device-driven Krylov loop already converged (uniform; nothing written)
    for (int r = r0 + tid; __builtin_amdgcn_ballot_w64(r < r1) != 0; r += BLOCK) {      // wavefront-uniform trip count
        const bool mine = r < r1;
        const int pat = mine ? (int)rowpat[r] : 0;
        const int len = mine ? offL[pat * PATW_OFF + PATW_LEN] : 0;
        const unsigned rb8 = (unsigned)r * 8u;            // 32-bit byte offsets on a scalar base (the plan checks n + max offset < 2^29)
        double wr = 0.0, acc = acc0;
        if (DOT >= 1 && mine) wr = wdot[r];
        for (int c = 0; __builtin_amdgcn_ballot_w64(c < len) != 0; c += 8) {            // (uniform) 8 entries at a time
            if (c < len) {
                const v4i32 oa = *reinterpret_cast<const v4i32 *>(offL + pat * PATW_OFF + c), ob = *reinterpret_cast<const v4i32 *>(offL + pat * PATW_OFF + c + 4);
                const int o[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
                double xx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) xx[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)o[u]));
                const v2f64 *q = reinterpret_cast<const v2f64 *>(valL + pat * PATW_LEN + c);
                const v2f64 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const double v[8] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
#pragma unroll
                for (int u = 0; u < 8; u++) { const double t = v[u] * xx[u]; acc += (c + u < len) ? t : -0.0; }     // -0.0 terms leave any sum bit-unchanged
            }
        }
        if (mine) {
            store_stream(reinterpret_cast<double *>(reinterpret_cast<char *>(y) + rb8), acc);
            dots.add_loaded(wr, acc);
        }
    }
    publish_dots<BLOCK, DOT>(dots, dot_scratch, partial, lb, pstride ? pstride : nb);
}

// The same with x STAGED and the dominant pattern in scalar registers -- the constant-coefficient 27-point stencil (the matrix of the reference's
// spmvtest3b and of HPCG), the 9- and 19-point ones.  The kernel above issues a gather per entry (27 vector-memory instructions per 64 rows, and
// every workgroup first copies all records into LDS); what such a kernel pays for is the number of those instructions
// (profiles/r03_pattern_team_kernel.txt).  Here a wavefront owns 64 consecutive rows, one per lane.  The x values they need from a run of m
// neighbouring columns of the dominant pattern are 63 + m consecutive doubles: all runs together (594 doubles for the 27-point stencil) come in
// by ceil(slots / 128) coalesced 16 B loads whose column offsets are a per-lane table (two loads), speculative addresses clamped to the array, and
// land in the wavefront's own LDS slice -- no barrier.  A row on the dominant pattern then multiplies slot (entry) + lane by the entry's value, a
// SCALAR register (the pattern's slots and values are kernel arguments); rows on other patterns whose offsets the dominant one's runs hold take
// their mask and values in the dominant pattern's slots by scalar loads, one round per distinct pattern among the lanes (dom_waterfall's scheme);
// foreign rows walk their own record.  Same products in the same order -- masked slots add -0.0 -- so y is the reference's, bit for bit.
struct WideDom { int len, pat, slots, maxcol; int slot[PATW_LEN]; double val[PATW_LEN]; int tri, pad[3]; };      // tri: entries 3q, 3q+1, 3q+2 sit in three neighbouring slots for every q (box stencils)
constexpr int WREC = 40;                                   // doubles per pattern in wdrec: 32 values in the dominant pattern's slots, [32] = mask | foreign << 32
template <int BLOCK, int NL, int CH, int DOT = 0>
__global__ __launch_bounds__(BLOCK)
void spmv_csr_valuerecw_staged_kernel(const unsigned char *__restrict__ rowpat, const double *__restrict__ wdrec, const v4i32 *__restrict__ wstage,
                                      const v4i32 *__restrict__ rec, int npat, const double *__restrict__ x, double *__restrict__ y, Rows RW,
                                      const WideDom D, int xcap, const double *__restrict__ guard = nullptr,
                                      const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr, int pstride = 0, int xs_plane = 0)
{
    if (guard != nullptr && guard[0] != 0.0) return;               // (fused forms) device-driven Krylov loop already converged
    extern __shared__ __attribute__((aligned(16))) double wide_dyn[];
    // XCD strips (round 5; xcd_strip_unit): workgroup -> chunk of 256 rows.  This kernel streams one byte per row, so x IS its traffic, and in the natural order the
    // x a chunk stages (its own rows, the lines next to them, the planes before and behind) crosses the fabric once per XCD that touches it: 5.9 x at 200^3
    // (profiles/r03_wide_records_staged.txt).  With strips an XCD keeps its eighth of every plane and finds the planes before and behind in its own L2.  The
    // partial sums of the fused dots stay with the CHUNK (not with the workgroup that ran it): the fold's order does not move.
    const int wg = xcd_strip_unit((int)blockIdx.x, (int)gridDim.x, xs_plane);
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    if (DOT != 0) {                                                // the only barrier: at the start (workgroup_dots_last)
        if (threadIdx.x == 0) dot_count = 0u;
        __syncthreads();
    }
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (WAVE - 1);
    double *xL = wide_dyn + w * xcap;
    // a wavefront walks CH consecutive chunks of 64 rows: the staging loads and the pattern byte of the NEXT chunk are issued before the products of the
    // one in hand are formed, so that a round trip hides behind the arithmetic (and the offset table is read once)
    const int rbase = RW.rb + (wg * (BLOCK / WAVE) + w) * (CH * WAVE);
    if (DOT == 0 && rbase >= RW.re) return;
    double c0 = 0.0, c1 = 0.0;                                     // (DOT) this lane's <w, y> and <y, y>: one partial per workgroup, folded by the caller
    if (rbase < RW.re) {
    const v4i32 so0 = wstage[2 * lane], so1 = wstage[2 * lane + 1];      // the column offsets of this lane's slot pairs, load by load
    const int so[8] = {so0.x, so0.y, so0.z, so0.w, so1.x, so1.y, so1.z, so1.w};
    const double *xr = xL + lane;
    v2f64 xs[NL];
    int patn;
    double wn = 0.0;
    auto prefetch = [&](int r0) {
        patn = rowpat[min(r0 + lane, RW.re - 1)];
        if (DOT != 0) wn = wdot[min(r0 + lane, RW.re - 1)];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int c = r0 + so[k], cc = min(max(c, 0), D.maxcol - 1);
            const v2f64 v = *reinterpret_cast<const v2f64u *>(x + cc);
            xs[k].x = c > cc ? v.y : v.x;                          // a pair pushed inside the array by the clamp hands each slot the half that holds its column
            xs[k].y = c < cc ? v.x : v.y;
        }
    };
    prefetch(rbase);
#pragma unroll 1
    for (int ch = 0; ch < CH; ch++) {
        const int r0 = rbase + ch * WAVE, r1 = min(r0 + WAVE, RW.re);
        if (r0 >= r1) break;                                        // (uniform)
        const int r = min(r0 + lane, r1 - 1);
        const bool live = r0 + lane < r1;
        __builtin_amdgcn_s_waitcnt(0);                              // the chunk's x and pattern bytes have landed (and the products before have read the slice)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NL; k++) { const int sl = 2 * (k * WAVE + lane); if (sl < D.slots) *reinterpret_cast<v2f64 *>(xL + sl) = xs[k]; }
        const int pat = patn;
        const double wv = wn;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ch + 1 < CH && r0 + WAVE < RW.re) prefetch(r0 + WAVE);  // (uniform)
        double acc = RW.acc0;
        if (__builtin_amdgcn_ballot_w64(live && pat != D.pat) == 0) {   // (uniform) every row here is on the dominant pattern
            if (D.tri) {                                            // three neighbouring slots per triple: one 16 B + one 8 B LDS read and one address for three entries
#pragma unroll
                for (int q = 0; q < PATW_LEN / 3; q++)
                    if (3 * q < D.len) {
                        const double *xp = xr + D.slot[3 * q];
                        const v2f64 x01 = *reinterpret_cast<const v2f64u *>(xp);
                        const double x2 = xp[2];
                        acc += D.val[3 * q] * x01.x; acc += D.val[3 * q + 1] * x01.y; acc += D.val[3 * q + 2] * x2;
                    }
            } else {
#pragma unroll
            for (int j = 0; j < PATW_LEN; j++) if (j < D.len) acc += D.val[j] * xr[D.slot[j]];
            }
        } else {
            // masks first: one 8 B scalar load per distinct pattern among the lanes.  Bit 32: foreign (the row walks its own record); bit 33: the pattern's values are
            // the dominant one's in every slot it keeps (HPCG's boundary rows: 26 and -1 everywhere) -- then the values stay in scalar registers
            unsigned m = D.len >= 32 ? 0xffffffffu : ((1u << D.len) - 1u);
            bool foreign = false, differs = false;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(live && pat != D.pat);
            while (todo != 0) {                                         // (uniform)
                const int q = __builtin_amdgcn_readlane(pat, __builtin_ctzll(todo));
                const unsigned long long bits = (unsigned long long)__double_as_longlong(wdrec[(size_t)q * WREC + 32]);      // (uniform address)
                const bool me = live && pat == q;
                m = me ? (unsigned)bits : m;
                foreign = me ? ((bits >> 32) & 1ull) != 0 : foreign;
                differs = me ? ((bits >> 33) & 1ull) == 0 : differs;
                todo &= ~__builtin_amdgcn_ballot_w64(me);
            }
            if (__builtin_amdgcn_ballot_w64(differs && !foreign) == 0) {      // (uniform) nobody needs values of its own
                if (!foreign) {
                    if (D.tri) {
#pragma unroll
                        for (int q = 0; q < PATW_LEN / 3; q++)
                            if (3 * q < D.len) {
                                const double *xp = xr + D.slot[3 * q];
                                const v2f64 x01 = *reinterpret_cast<const v2f64u *>(xp);
                                const double x2 = xp[2];
                                const double t0 = D.val[3 * q] * x01.x, t1 = D.val[3 * q + 1] * x01.y, t2 = D.val[3 * q + 2] * x2;
                                acc += ((m >> (3 * q)) & 1u) ? t0 : -0.0; acc += ((m >> (3 * q + 1)) & 1u) ? t1 : -0.0; acc += ((m >> (3 * q + 2)) & 1u) ? t2 : -0.0;
                            }
                    } else {
#pragma unroll
                    for (int j = 0; j < PATW_LEN; j++) if (j < D.len) { const double t = D.val[j] * xr[D.slot[j]]; acc += ((m >> j) & 1u) ? t : -0.0; }     // -0.0 terms leave any sum bit-unchanged
                    }
                }
            } else {
                // ... in two halves of 16 slots (32 registers of values at a time keep the kernel at 64 registers: eight wavefronts per SIMD)
#pragma unroll
                for (int h = 0; h < PATW_LEN; h += 16) {
                    if (h < D.len) {                                    // (uniform)
                        double v[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) v[j] = D.val[h + j];
                        todo = __builtin_amdgcn_ballot_w64(differs && !foreign);
                        while (todo != 0) {                             // (uniform) the values in the dominant pattern's slots, by scalar loads
                            const int q = __builtin_amdgcn_readlane(pat, __builtin_ctzll(todo));
                            const double *R = wdrec + (size_t)q * WREC + h;
                            const bool me = live && pat == q;
#pragma unroll
                            for (int j = 0; j < 16; j++) v[j] = me ? R[j] : v[j];
                            todo &= ~__builtin_amdgcn_ballot_w64(me);
                        }
                        if (!foreign) {
#pragma unroll
                            for (int j = 0; j < 16; j++) if (h + j < D.len) { const double t = v[j] * xr[D.slot[h + j]]; acc += ((m >> (h + j)) & 1u) ? t : -0.0; }
                        }
                    }
                }
            }
            if (foreign) {                                              // a pattern the dominant one's runs do not hold: its own record (offsets, values), entry by entry
                const int *off = reinterpret_cast<const int *>(rec) + pat * PATW_OFF;
                const double *vv = reinterpret_cast<const double *>(rec + npat * (PATW_OFF / 4)) + pat * PATW_LEN;
                const int len = off[PATW_LEN];
                const unsigned rb8 = (unsigned)r * 8u;
                for (int j = 0; j < len; j++) acc += vv[j] * *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)off[j]));
            }
        }
        if (live) store_stream(y + r, acc);
        if (DOT >= 1 && live) c0 += wv * acc;
        if (DOT >= 2 && live) c1 += acc * acc;
    }
    }
    if (DOT != 0) workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, wg, pstride ? pstride : (int)gridDim.x, true);
}

// BLOCK ROWS: the row form of a b x b blocked stencil (liship_bsr_to_rows: CSR rows that list lis_matvec_bsr's terms) has b interior row patterns that take
// turns, one per place in the block -- but the b rows of a block row list the SAME columns in the same order (the block row's blocks, column after column),
// only their values differ.  Here a LANE owns a block row: it reads each x of the block row once from the wavefront's staged window (64 block rows, the
// dominant block row's runs of columns, 63 b + m doubles each) and feeds its b running sums, entry after entry in the rows' own order, the values b x len
// scalars of the kernel's arguments.  A block row is named by the pattern byte of its FIRST row (plan time checks that it determines the others'); block rows
// whose columns are a subsequence of the dominant one's with its values in the entries they keep are a mask (-0.0 terms), the rest walk their rows' own wide
// records.  14 x reads and 28 multiply-adds for the two rows of a 2 x 2 block row where the row-by-row kernel on the virtual pattern spends 40 masked terms.
// Same products in the same order as lis_matvec_bsr.c:293-343 (explicit zeros included): the reference's bits.
struct BlockDom { int len, key, slots, maxcol, b, pair, pad0, pad1; int slot[PATW_LEN]; double val[4][PATW_LEN]; };
template <int B, int NL, int DOT = 0>
__global__ __launch_bounds__(256)
void spmv_csr_blockrows_staged_kernel(const unsigned char *__restrict__ rowpat, const unsigned long long *__restrict__ bdrec, const int *__restrict__ bstage,
                                      const v4i32 *__restrict__ rec, int npat, const double *__restrict__ x, double *__restrict__ y, Rows RW,
                                      const BlockDom D, int xcap, const double *__restrict__ guard = nullptr,
                                      const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr, int pstride = 0, int xs_plane = 0)
{
    constexpr int BLOCK = 256;
    if (guard != nullptr && guard[0] != 0.0) return;               // (fused forms) device-driven Krylov loop already converged
    extern __shared__ __attribute__((aligned(16))) double wide_dyn[];
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    if (DOT != 0) {                                                // the only barrier: at the start (workgroup_dots_last)
        if (threadIdx.x == 0) dot_count = 0u;
        __syncthreads();
    }
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (WAVE - 1);
    double *xL = wide_dyn + w * xcap;
    const int wg = xcd_strip_unit((int)blockIdx.x, (int)gridDim.x, xs_plane);      // (XCD strips: see spmv_csr_valuerecw_staged_kernel)
    const int r0 = RW.rb + (wg * (BLOCK / WAVE) + w) * (WAVE * B);      // (RW.rb and RW.re are multiples of B: the launcher checks)
    double c0 = 0.0, c1 = 0.0;
    if (r0 < RW.re) {
        const int rl = r0 + lane * B;
        const bool live = rl < RW.re;
        const int key = live ? (int)rowpat[rl] : D.key;
        v2f64 xs[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int c = r0 + bstage[k * WAVE + lane], cc = min(max(c, 0), D.maxcol - 1);
            const v2f64 v = *reinterpret_cast<const v2f64u *>(x + cc);
            xs[k].x = c > cc ? v.y : v.x;                          // a pair pushed inside the array by the clamp hands each slot the half that holds its column
            xs[k].y = c < cc ? v.x : v.y;
        }
        double wv[B];
        if (DOT != 0) {
#pragma unroll
            for (int k = 0; k < B; k++) wv[k] = live ? wdot[rl + k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) { const int sl = 2 * (k * WAVE + lane); if (sl < D.slots) *reinterpret_cast<v2f64 *>(xL + sl) = xs[k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *xr = xL + lane * B;
        double acc[B];
#pragma unroll
        for (int k = 0; k < B; k++) acc[k] = RW.acc0;
        if (__builtin_amdgcn_ballot_w64(key != D.key) == 0) {      // (uniform) every block row here is the dominant one
            if (D.pair) {                                          // entries 2q, 2q + 1 sit in neighbouring slots (even b: a block's columns two by two)
#pragma unroll
                for (int q = 0; q < PATW_LEN / 2; q++)
                    if (2 * q < D.len) {
                        const v2f64 x01 = *reinterpret_cast<const v2f64u *>(xr + D.slot[2 * q]);
#pragma unroll
                        for (int k = 0; k < B; k++) { acc[k] += D.val[k][2 * q] * x01.x; acc[k] += D.val[k][2 * q + 1] * x01.y; }
                    }
            } else {
#pragma unroll
                for (int j = 0; j < PATW_LEN; j++)
                    if (j < D.len) {
                        const double xj = xr[D.slot[j]];
#pragma unroll
                        for (int k = 0; k < B; k++) acc[k] += D.val[k][j] * xj;
                    }
            }
        } else {
            // masks: one 8 B scalar load per distinct key among the lanes.  Bit 32: foreign (the block row's rows walk their own records)
            unsigned m = D.len >= 32 ? 0xffffffffu : ((1u << D.len) - 1u);
            bool foreign = false;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(key != D.key);
            while (todo != 0) {                                     // (uniform)
                const int q = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
                const unsigned long long bits = bdrec[q];           // (uniform address)
                const bool me = key == q;
                m = me ? (unsigned)bits : m;
                foreign = me ? ((bits >> 32) & 1ull) != 0 : foreign;
                todo &= ~__builtin_amdgcn_ballot_w64(me);
            }
            if (!foreign) {
#pragma unroll
                for (int j = 0; j < PATW_LEN; j++)
                    if (j < D.len) {
                        const double xj = xr[D.slot[j]];
                        const bool keep = ((m >> j) & 1u) != 0;
#pragma unroll
                        for (int k = 0; k < B; k++) { const double t = D.val[k][j] * xj; acc[k] += keep ? t : -0.0; }      // -0.0 terms leave any sum bit-unchanged
                    }
            } else if (live) {
#pragma unroll 1
                for (int k = 0; k < B; k++) {
                    const int r = rl + k, pat = (int)rowpat[r];
                    const int *off = reinterpret_cast<const int *>(rec) + pat * PATW_OFF;
                    const double *vv = reinterpret_cast<const double *>(rec + npat * (PATW_OFF / 4)) + pat * PATW_LEN;
                    const int len = off[PATW_LEN];
                    const unsigned rb8 = (unsigned)r * 8u;
                    double a = RW.acc0;
                    for (int j = 0; j < len; j++) a += vv[j] * *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (rb8 + (unsigned)off[j]));
                    acc[k] = a;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < B; k++) store_stream(y + rl + k, acc[k]);
        }
        if (DOT != 0 && live) {
#pragma unroll
            for (int k = 0; k < B; k++) { c0 += wv[k] * acc[k]; if (DOT >= 2) c1 += acc[k] * acc[k]; }
        }
    }
    if (DOT != 0) workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, wg, pstride ? pstride : (int)gridDim.x, true);
}

// Z-MARCHING form of the 27-point box stencil with constant coefficients (round 5): the matrix of the reference's spmvtest3b (test/spmvtest3b.c:136-160) and of
// HPCG, rows in ascending column order, on a grid that is a BOX -- plan time has checked, row by row, that a row lacks exactly the neighbours that lie outside the
// grid (wide_box_check) and that every pattern carries the dominant pattern's values.  The staged kernel above brings the x of a wavefront's 64 rows in by nine
// runs of loads through L1 and runs at a quarter of the roofline: it is bound by the chain of round trips inside a wavefront, not by bytes.  Here a workgroup owns a
// tile of 128 columns x TY lines and WALKS the planes, as the 7-point marching kernel does: a plane's tile (+ a halo line above and below, a halo column left and
// right, the four corners) is loaded ONCE by coalesced 16 B loads issued D planes ahead, parked in registers, written to one of two LDS buffers; a lane takes the
// LPW + 2 lines x 4 columns its two rows per line need from the NEWEST plane out of LDS -- (LPW + 2) x (16 + 8 + 8) B per plane instead of 27 entries per row --
// and keeps the two planes before it in registers.  The 27 values are kernel arguments (scalar registers).  Same products in the same order, one rounded multiply
// and one rounded add per term: the reference's bits.  A neighbour outside the grid is a halo cell that holds a ZERO: its term is +-0.0, and a sum that starts at
// +0.0 is never -0.0, so the term leaves every bit where a missing slot leaves it: no masks, no pattern bytes -- x once and y once, 16 B per row.
struct Box27 { int S, SO, tiles_x, tiles_y, zseg, nseg, z0, z1, wgs, xcd, planes, pad; double poison; double val[27]; };
template <int LPW, int DOT, bool WS, bool PT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LPW == 2 ? (WS ? 2 : 3) : (WS ? 3 : 4))))      // (eight lines per tile: 171 registers without the hint -- three short of a third workgroup per CU; with a vector w of their own the forms spilled 140 registers under that cap: one workgroup fewer per CU instead)
void spmv_csr_box27_march_kernel(const double *__restrict__ x, double *__restrict__ y, double acc0, const Box27 M, int nx,
                                 const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                                 const double *__restrict__ guard = nullptr, int pstride = 0)
{
    constexpr int BLOCK = 256, TX = 128, TY = 4 * LPW, LX = TX + 4, NLN = LPW + 2;      // an LDS line: [pad][left halo][TX columns][right halo][pad]: a lane's pair at 2 + 2 lane, 16 B aligned
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;      // device-driven Krylov loop already converged (nothing has been written)
    __shared__ __attribute__((aligned(16))) double buf[2][(TY + 2) * LX];
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    const int tid = (int)threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & (WAVE - 1);
    if (DOT != 0 && tid == 0) dot_count = 0u;                         // (a barrier follows before anyone counts itself in)
    int wg = (int)blockIdx.x;
    const int ntile = M.tiles_x * M.tiles_y;
    if (M.xcd) { const int e = M.wgs / NUM_XCD; if (wg < e * NUM_XCD) { const int k = wg % NUM_XCD, j = wg / NUM_XCD; wg = k * e + j; } }      // each XCD a contiguous eighth of the (segment, tile) list
    const int seg = wg / ntile, t = wg - seg * ntile;
    const int ty = t / M.tiles_x, tx = t - ty * M.tiles_x;
    const int za = M.z0 + seg * M.zseg, zb = min(M.z1, za + M.zseg);
    const int col0 = tx * TX, line0 = ty * TY;
    const long long S = M.S, SO = M.SO;
    // partial tiles (round 5, as in spmv_csr_valuerec_march_kernel): the last tile of a line holds SX < 128 columns (even, >= 4), the last tile of a plane LY < TY lines.  Lanes /
    // lines beyond them repeat the last pair's / line's loads, park nothing, store nothing; the right halo column sits behind the last pair, the bottom halo line behind the last line.
    // PT = false (grids of whole tiles) folds all of it away: the partial-tile tests in the plane loop cost the whole-tile grids a third of their rate (256^3: 0.060 -> 0.080 ms).
    const int SX = PT ? min(TX, M.S - col0) : TX, hlane = PT ? (SX >> 1) - 1 : WAVE - 1, lc = PT ? 2 * min(lane, hlane) : 2 * lane;
    const int LY = PT ? min(TY, (int)(SO / S) - line0) : TY;
    const bool act = PT ? lane <= hlane : true;
    auto lval = [&](int i) { return PT ? w * LPW + i < LY : true; };
    auto lidx = [&](int i) { return PT ? line0 + min(w * LPW + i, LY - 1) : line0 + w * LPW + i; };
    auto roff = [&](int i) { return (long long)lidx(i) * S + col0 + lc; };      // this lane's pair of line i inside a plane
    auto at = [&](long long a) { return (int)(a < 0 ? 0 : (a > (long long)nx - 2 ? (long long)nx - 2 : a)); };      // (pairs: the last start is nx - 2; clamped addresses only ever feed poisoned cells)
    const double pz = M.poison;
    const bool box_left = tx == 0, box_right = tx == M.tiles_x - 1, box_top = ty == 0, box_bottom = ty == M.tiles_y - 1;      // (uniform)
    struct Packet { v2f64 own[LPW]; v2f64 hy; double hx[LPW]; double hc; v2f64 ww[WS ? LPW : 1]; };
    auto load_packet = [&](Packet &P, int z, bool with_w) {
        const long long zo = (long long)z * SO;
#pragma unroll
        for (int i = 0; i < LPW; i++) P.own[i] = *reinterpret_cast<const v2f64u *>(x + at(zo + roff(i)));
        const long long hl = zo + (long long)(w == 0 ? line0 - 1 : line0 + LY) * S + col0;      // the halo line this wavefront brings (the first and the last wavefront)
        if (w == 0 || w == 3) P.hy = *reinterpret_cast<const v2f64u *>(x + at(hl + lc));
        if (lane == 0 || lane == hlane) {
            const int hcol = lane == 0 ? col0 - 1 : col0 + SX;
#pragma unroll
            for (int i = 0; i < LPW; i++) { const long long a = zo + (long long)lidx(i) * S + hcol; P.hx[i] = x[a < 0 ? 0 : (a > (long long)nx - 1 ? (long long)nx - 1 : a)]; }
            if (w == 0 || w == 3) { const long long a = zo + (long long)(w == 0 ? line0 - 1 : line0 + LY) * S + hcol; P.hc = x[a < 0 ? 0 : (a > (long long)nx - 1 ? (long long)nx - 1 : a)]; }
        }
        if (DOT != 0 && WS && with_w && z < M.z1) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.ww[i] = *reinterpret_cast<const v2f64u *>(wdot + zo + roff(i));
        }
    };
    // a plane into an LDS buffer: the cells outside the grid take the zero (a whole plane of it before the first and behind the last plane)
    auto store_packet = [&](const Packet &P, double *B, int z) {
        const bool off = z < 0 || z >= M.planes;                      // (uniform)
#pragma unroll
        for (int i = 0; i < LPW; i++) if (act && lval(i)) { v2f64 v = P.own[i]; if (off) { v.x = v.y = pz; } *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + 2 + 2 * lane) = v; }
        if (w == 0 && act) { v2f64 h = P.hy; if (off || box_top) { h.x = h.y = pz; } *reinterpret_cast<v2f64 *>(B + 2 + 2 * lane) = h; }
        if (w == 3 && act) { v2f64 h = P.hy; if (off || box_bottom) { h.x = h.y = pz; } *reinterpret_cast<v2f64 *>(B + (LY + 1) * LX + 2 + 2 * lane) = h; }
        if (lane == 0 || lane == hlane) {
            const int c = lane == 0 ? 1 : 2 + SX;
            // (a tile one pair wide cannot be: SX >= 4, so lane 0 and lane hlane are two lanes and each writes one side)
            const bool side = lane == 0 ? box_left : box_right;
#pragma unroll
            for (int i = 0; i < LPW; i++) if (lval(i)) B[(w * LPW + i + 1) * LX + c] = (off || side) ? pz : P.hx[i];
            if (w == 0) B[c] = (off || side || box_top) ? pz : P.hc;
            if (w == 3) B[(LY + 1) * LX + c] = (off || side || box_bottom) ? pz : P.hc;
        }
    };
    // the lines a lane's rows need from a plane in LDS: LPW + 2 of them, four columns each (left, its pair, right)
    auto read_plane = [&](double (&X)[NLN][4], const double *B) {
#pragma unroll
        for (int j = 0; j < NLN; j++) {
            const int li = (w * LPW + j) * LX + 2 + 2 * lane;
            const v2f64 c = *reinterpret_cast<const v2f64 *>(B + li);
            X[j][0] = B[li - 1]; X[j][1] = c.x; X[j][2] = c.y; X[j][3] = B[li + 2];
        }
    };
    double Xa[NLN][4], Xb[NLN][4], Xc[NLN][4];
    Packet Q0, Q1;
    v2f64 ww0[WS ? LPW : 1];
    {   // prologue: plane za - 1 and plane za through the two buffers into registers; planes za + 1, za + 2 in flight
        Packet P;
        load_packet(P, za - 1, false);
        store_packet(P, buf[0], za - 1);
        load_packet(P, za, true);
        store_packet(P, buf[1], za);
        if (WS) {
#pragma unroll
            for (int i = 0; i < LPW; i++) ww0[i] = P.ww[i];
        }
        load_packet(Q0, za + 1, true);
        load_packet(Q1, za + 2, true);
        __syncthreads();
        read_plane(Xa, buf[0]);
        read_plane(Xb, buf[1]);
        __syncthreads();                                              // both buffers are free again
    }
    double c0 = 0.0, c1 = 0.0;
    // one plane: Xp / Xc hold planes z - 1 / z, packet P plane z + 1 (the oldest in flight), which goes to LDS buffer (z + 1) & 1 and from there into Xn
    auto step = [&](int z, double (&Xp)[NLN][4], double (&Xq)[NLN][4], double (&Xn)[NLN][4], Packet &P) {
        double *B = buf[(z + 1) & 1];
        store_packet(P, B, z + 1);
        v2f64 wnext[WS ? LPW : 1];
        if (WS) {
#pragma unroll
            for (int i = 0; i < LPW; i++) wnext[i] = P.ww[i];
        }
        load_packet(P, z + 3, true);                                  // this slot's next plane
        __syncthreads();                                              // plane z + 1 is in LDS (and nobody reads the other buffer any more)
        read_plane(Xn, B);
#pragma unroll
        for (int i = 0; i < LPW; i++) {
            double s0 = acc0, s1 = acc0;
#pragma unroll
            for (int dz = 0; dz < 3; dz++) {
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const double *L = dz == 0 ? Xp[i + dy] : dz == 1 ? Xq[i + dy] : Xn[i + dy];
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const double v = M.val[dz * 9 + dy * 3 + dx];
                        s0 += v * L[dx];
                        s1 += v * L[dx + 1];
                    }
                }
            }
            const long long row = (long long)z * SO + roff(i);
            const bool live = act && lval(i);
            v2f64 out; out.x = s0; out.y = s1;
            if (live) store_stream(reinterpret_cast<v2f64 *>(y + row), out);
            if (DOT != 0 && live) {
                v2f64 wv;
                if (WS) wv = ww0[i]; else { wv.x = Xq[i + 1][1]; wv.y = Xq[i + 1][2]; }
                c0 += wv.x * s0; c0 += wv.y * s1;
                if (DOT >= 2) { c1 += s0 * s0; c1 += s1 * s1; }
            }
        }
        if (WS) {
#pragma unroll
            for (int i = 0; i < LPW; i++) ww0[i] = wnext[i];
        }
    };
    for (int z = za; z < zb; z += 6) {                                // the three register planes and the two packets rotate by NAME: six steps per trip, no moves
        step(z, Xa, Xb, Xc, Q0);
        if (z + 1 < zb) step(z + 1, Xb, Xc, Xa, Q1);
        if (z + 2 < zb) step(z + 2, Xc, Xa, Xb, Q0);
        if (z + 3 < zb) step(z + 3, Xa, Xb, Xc, Q1);
        if (z + 4 < zb) step(z + 4, Xb, Xc, Xa, Q0);
        if (z + 5 < zb) step(z + 5, Xc, Xa, Xb, Q1);
    }
    if (DOT != 0) workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, (int)blockIdx.x, pstride ? pstride : M.wgs, true);
}

// Z-MARCHING form of the block-row product for the 7-point stencil kept as 2 x 2 BLOCKS (round 5): Lis's default BSR block size on its own test matrices
// (`-storage bsr`, lis_matvec_bsr.c:293-343), constant coefficients, the grid a box.  A lane's pair of rows IS a block row; block j of it brings the pair
// (x[2 bj], x[2 bj + 1]) and both rows add value * x column by column, block after block in the order the conversion met them, explicit zeros included -- 14 terms
// per row.  The seven pairs a block row needs are the 7-point marching kernel's: the planes before and behind in registers, the lines above and below and the
// blocks left and right from the LDS copy of the plane in hand (two halo columns on either side).  The term list (which pair, which column, the two rows' values)
// rides as kernel arguments; a block outside the grid is a pair of zeros in the halo, whose terms (+-0.0) cannot change a sum that started at +0.0.
// ORD: the order of the seven blocks in a block row -- 0 ascending columns (-SO, -S, left, own, right, +S, +SO); 1 and 2 the orders lis_matrix_convert_csr2bsr meets
// them in (row after row of the block row, lis_matrix_bsr.c:351-552) when the CSR rows are sorted (-SO, -S, left, own, +S, +SO, right) or in the order of the
// reference's generators (test/test3.c:114-127: -SO, +SO, -S, +S, left, own, right).  Compile-time, so that the 14 terms of a row are straight-line
// code on registers (a run-time term list cost 350 branches and spilled the scalar registers: 0.108 ms at 256^3 against the staged kernel's 0.08).
struct Block2March { int S, SO, tiles_x, tiles_y, zseg, nseg, z0, z1, wgs, xcd, planes, ord; double v0[14], v1[14]; };
template <int ORD, int DOT, bool WS, bool PT = true>
__global__ __launch_bounds__(256)
void spmv_csr_block2_march_kernel(const double *__restrict__ x, double *__restrict__ y, const Block2March M, int nx,
                                  const double *__restrict__ wdot = nullptr, double *__restrict__ partial = nullptr,
                                  const double *__restrict__ guard = nullptr, int pstride = 0)
{
    constexpr int BLOCK = 256, LPW = 2, TX = 128, TY = 4 * LPW, LX = TX + 4;      // an LDS line: [2 left halo][TX columns][2 right halo]: a lane's pair at 2 + 2 lane, 16 B aligned
    if (DOT != 0 && guard != nullptr && guard[0] != 0.0) return;      // device-driven Krylov loop already converged
    __shared__ __attribute__((aligned(16))) double buf[2][(TY + 2) * LX];
    __shared__ double dot_part[(DOT >= 2 ? 2 : 1) * (DOT != 0 ? BLOCK : 1)];
    __shared__ unsigned dot_count;
    const int tid = (int)threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & (WAVE - 1);
    if (DOT != 0 && tid == 0) dot_count = 0u;
    int wg = (int)blockIdx.x;
    const int ntile = M.tiles_x * M.tiles_y;
    if (M.xcd) { const int e = M.wgs / NUM_XCD; if (wg < e * NUM_XCD) { const int k = wg % NUM_XCD, j = wg / NUM_XCD; wg = k * e + j; } }
    const int seg = wg / ntile, t = wg - seg * ntile;
    const int ty = t / M.tiles_x, tx = t - ty * M.tiles_x;
    const int za = M.z0 + seg * M.zseg, zb = min(M.z1, za + M.zseg);
    const int col0 = tx * TX, line0 = ty * TY;
    const long long S = M.S, SO = M.SO;
    // partial tiles (round 5, as in spmv_csr_valuerec_march_kernel): SX < 128 columns in the last tile of a line (whole blocks: even, >= 4), LY < TY lines in the last tile of a plane
    // PT = false (grids of whole tiles) folds these tests away: they cost 256^3 8 % (0.038 -> 0.041 ms).
    const int SX = PT ? min(TX, M.S - col0) : TX, hlane = PT ? (SX >> 1) - 1 : WAVE - 1, lc = PT ? 2 * min(lane, hlane) : 2 * lane;
    const int LY = PT ? min(TY, (int)(SO / S) - line0) : TY;
    const bool act = PT ? lane <= hlane : true;
    auto lval = [&](int i) { return PT ? w * LPW + i < LY : true; };
    auto lidx = [&](int i) { return PT ? line0 + min(w * LPW + i, LY - 1) : line0 + w * LPW + i; };
    auto roff = [&](int i) { return (long long)lidx(i) * S + col0 + lc; };
    auto at = [&](long long a) { return (int)(a < 0 ? 0 : (a > (long long)nx - 2 ? (long long)nx - 2 : a)); };      // (clamped addresses only ever feed zeroed cells)
    const bool box_left = tx == 0, box_right = tx == M.tiles_x - 1, box_top = ty == 0, box_bottom = ty == M.tiles_y - 1;      // (uniform)
    v2f64 zero2; zero2.x = 0.0; zero2.y = 0.0;
    struct Packet { v2f64 own[LPW]; v2f64 hy; v2f64 hx[LPW]; v2f64 ww[WS ? LPW : 1]; };
    auto load_packet = [&](Packet &P, int z, bool with_w) {
        const long long zo = (long long)z * SO;
#pragma unroll
        for (int i = 0; i < LPW; i++) P.own[i] = *reinterpret_cast<const v2f64u *>(x + at(zo + roff(i)));
        if (w == 0 || w == 3) P.hy = *reinterpret_cast<const v2f64u *>(x + at(zo + (long long)(w == 0 ? line0 - 1 : line0 + LY) * S + col0 + lc));
        if (lane == 0 || lane == hlane) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.hx[i] = *reinterpret_cast<const v2f64u *>(x + at(zo + (long long)lidx(i) * S + (lane == 0 ? col0 - 2 : col0 + SX)));
        }
        if (DOT != 0 && WS && with_w && z < M.z1) {
#pragma unroll
            for (int i = 0; i < LPW; i++) P.ww[i] = *reinterpret_cast<const v2f64u *>(wdot + zo + roff(i));
        }
    };
    auto store_packet = [&](const Packet &P, double *B) {
#pragma unroll
        for (int i = 0; i < LPW; i++) if (act && lval(i)) *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + 2 + 2 * lane) = P.own[i];
        if (w == 0 && act) *reinterpret_cast<v2f64 *>(B + 2 + 2 * lane) = box_top ? zero2 : P.hy;
        if (w == 3 && act) *reinterpret_cast<v2f64 *>(B + (LY + 1) * LX + 2 + 2 * lane) = box_bottom ? zero2 : P.hy;
        if (lane == 0 || lane == hlane) {
            const bool side = lane == 0 ? box_left : box_right;
#pragma unroll
            for (int i = 0; i < LPW; i++) if (lval(i)) *reinterpret_cast<v2f64 *>(B + (w * LPW + i + 1) * LX + (lane == 0 ? 0 : 2 + SX)) = side ? zero2 : P.hx[i];
        }
    };
    Packet Q[2];
    v2f64 prev[LPW], ww0[WS ? LPW : 1];
    {
        Packet P;
        load_packet(P, za - 1, false);
#pragma unroll
        for (int i = 0; i < LPW; i++) prev[i] = za == 0 ? zero2 : P.own[i];      // (uniform) the grid's first plane has no plane before it
        load_packet(P, za, true);
        store_packet(P, buf[za & 1]);
        if (WS) {
#pragma unroll
            for (int i = 0; i < LPW; i++) ww0[i] = P.ww[i];
        }
    }
    load_packet(Q[0], za + 1, true);
    load_packet(Q[1], za + 2, true);
    __syncthreads();
    double c0 = 0.0, c1 = 0.0;
    for (int zq = za; zq < zb; zq += 2) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const int z = zq + d;
            if (z < zb) {                                             // (uniform)
                Packet &P = Q[d];                                     // plane z + 1: the oldest in flight
                store_packet(P, buf[(z + 1) & 1]);
                const double *B = buf[z & 1];
#pragma unroll
                for (int i = 0; i < LPW; i++) {
                    const int li = (w * LPW + i + 1) * LX + 2 + 2 * lane;
                    v2f64 src[7];
                    src[0] = prev[i];
                    src[1] = *reinterpret_cast<const v2f64 *>(B + li - LX);
                    src[2] = *reinterpret_cast<const v2f64 *>(B + li - 2);
                    src[3] = *reinterpret_cast<const v2f64 *>(B + li);
                    src[4] = *reinterpret_cast<const v2f64 *>(B + li + 2);
                    src[5] = *reinterpret_cast<const v2f64 *>(B + li + LX);
                    src[6] = z == M.planes - 1 ? zero2 : P.own[i];      // (uniform) the grid's last plane has no plane behind it
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int q = 0; q < 7; q++) {                     // block after block, column after column (lis_matvec_bsr.c:340-343)
                        constexpr int kinds[3][7] = {{0, 1, 2, 3, 4, 5, 6}, {0, 1, 2, 3, 5, 6, 4}, {0, 6, 1, 5, 2, 3, 4}};
                        const v2f64 pr = src[kinds[ORD][q]];
                        s0 += M.v0[2 * q] * pr.x; s1 += M.v1[2 * q] * pr.x;
                        s0 += M.v0[2 * q + 1] * pr.y; s1 += M.v1[2 * q + 1] * pr.y;
                    }
                    const long long row = (long long)z * SO + roff(i);
                    const bool live = act && lval(i);
                    v2f64 out; out.x = s0; out.y = s1;
                    if (live) store_stream(reinterpret_cast<v2f64 *>(y + row), out);
                    if (DOT != 0 && live) {
                        const v2f64 wv = WS ? ww0[i] : src[3];
                        c0 += wv.x * s0; c0 += wv.y * s1;
                        if (DOT >= 2) { c1 += s0 * s0; c1 += s1 * s1; }
                    }
                    prev[i] = src[3];
                    if (WS) ww0[i] = P.ww[i];
                }
                load_packet(Q[d], z + 3, true);
                __syncthreads();
            }
        }
    }
    if (DOT != 0) workgroup_dots_last<BLOCK, DOT>(c0, c1, dot_part, &dot_count, partial, (int)blockIdx.x, pstride ? pstride : M.wgs, true);
}

} // namespace
