// csr_plan.hpp -- the CSR plan of spmv_csr.hip (included there, once, behind csr_kernels.hpp): the plan-time kernels, struct liship_csr_plan_s and every
// liship_csr_plan_* / liship_spmv_csr_set_* entry point -- row split, index codes, row patterns, pattern / value records, dominant patterns and their box checks,
// block rows, block-local columns and their runs, the band scan.  Each step is optional and never an error when the matrix does not qualify (include/liship.h).
#pragma once
namespace {
// plan time: block row br = (z, y, xb) of a 2 x 2 blocked 7-point grid must keep exactly the terms of the dominant block row whose block lies inside the grid
// (bdrec[key]: a mask over the dominant block row's entries, bit 32 = foreign); term j's block is `kind[j]` (0 -SO, 1 -S, 2 left, 3 own, 4 right, 5 +S, 6 +SO)
__global__ void block2_box_check(int nbr, int S, int SO, int nterm, const unsigned char *__restrict__ rowpat, const unsigned long long *__restrict__ bdrec,
                                 const Block2March M, int *__restrict__ bad)
{
    const int br = blockIdx.x * blockDim.x + threadIdx.x;
    if (br >= nbr) return;
    const int r = 2 * br, z = r / SO, q = r - z * SO, yy = q / S, xx = q - yy * S, lines = SO / S, planes = M.planes;
    unsigned want = 0;
    const int kinds[3][7] = {{0, 1, 2, 3, 4, 5, 6}, {0, 1, 2, 3, 5, 6, 4}, {0, 6, 1, 5, 2, 3, 4}};
    for (int j = 0; j < nterm; j++) {
        const int k = kinds[M.ord][j >> 1];
        const bool in = k == 0 ? z > 0 : k == 1 ? yy > 0 : k == 2 ? xx > 0 : k == 3 ? true : k == 4 ? xx + 2 < S : k == 5 ? yy + 1 < lines : z + 1 < planes;
        if (in) want |= 1u << j;
    }
    const unsigned long long bits = bdrec[rowpat[r]];
    if ((bits >> 32) != 0ull || (unsigned)bits != want) atomicAdd(bad, 1);
}

// plan time: is the grid of a 27-point plan a BOX?  Row r = (z, y, x) must keep exactly the dominant pattern's slots whose neighbour (z + dz, y + dy, x + dx) lies
// inside the grid (masks: wdrec[pattern * WREC + 32], the low 27 bits).  bad[0] counts the rows that do not.
__global__ void wide_box_check(int n, int S, int SO, const unsigned char *__restrict__ rowpat, const double *__restrict__ wdrec, int *__restrict__ bad)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int z = r / SO, q = r - z * SO, yy = q / S, xx = q - yy * S, lines = SO / S, planes = n / SO;
    unsigned want = 0;
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++)
        if (z + dz >= 0 && z + dz < planes && yy + dy >= 0 && yy + dy < lines && xx + dx >= 0 && xx + dx < S) want |= 1u << ((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1));
    const unsigned long long bits = (unsigned long long)__double_as_longlong(wdrec[(size_t)rowpat[r] * WREC + 32]);
    if ((unsigned)(bits & 0x7ffffffull) != want || ((bits >> 32) & 3ull) != 2ull) atomicAdd(bad, 1);      // (bit 32: foreign; bit 33: the kept slots carry the dominant values)
}

// plan time: one 64-bit hash per row over (length, code sequence); the distinct ones in an open-addressing table together with the
// smallest row that has them; gives up beyond 255
constexpr int PAT_SLOTS = 1024, PAT_MAXLEN = 64;
__device__ __forceinline__ unsigned long long row_hash(const unsigned char *codes, int s, int e)
{
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)(e - s);
    for (int k = s; k < e; k++) { h ^= codes[k]; h *= 1099511628211ull; }
    return h | 1ull;                                 // 0 marks an empty slot
}
__global__ void csr_collect_patterns(int n, const int *__restrict__ ptr, const unsigned char *__restrict__ codes,
                                     unsigned long long *__restrict__ keys, int *__restrict__ rep, int *__restrict__ count)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || count[0] > 255) return;
    const int s = ptr[r], e = ptr[r + 1];
    if (e - s > PAT_MAXLEN) { atomicAdd(count, 1000); return; }
    const unsigned long long h = row_hash(codes, s, e);
    unsigned slot = (unsigned)(h >> 40) & (PAT_SLOTS - 1);
    for (int probe = 0; probe < PAT_SLOTS; probe++) {
        unsigned long long v = keys[slot];
        if (v == 0ull) {
            v = atomicCAS(&keys[slot], 0ull, h);
            if (v == 0ull) { atomicAdd(count, 1); v = h; }
        }
        if (v == h) { if (r < rep[slot]) atomicMin(&rep[slot], r); return; }    // the test keeps 10^8 rows off the same few words
        slot = (slot + 1) & (PAT_SLOTS - 1);
        if (count[0] > 255) return;
    }
}
// the code sequences of the representative rows: out[p * PAT_MAXLEN + j]
__global__ void csr_fetch_patterns(int npat, const int *__restrict__ rep, const int *__restrict__ ptr,
                                   const unsigned char *__restrict__ codes, int *__restrict__ len, unsigned char *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npat) return;
    const int s = ptr[rep[p]], e = ptr[rep[p] + 1];
    len[p] = e - s;
    for (int k = s; k < e && k - s < PAT_MAXLEN; k++) out[p * PAT_MAXLEN + (k - s)] = codes[k];
}
// one workgroup per row block: pattern number (position of the row's hash in the sorted list, sequence verified) and the row's
// start relative to the block; *bad counts rows that match no pattern (hash collision: the plan then stays with the codes)
__global__ void csr_encode_patterns(const v2i32 *__restrict__ blk, const int *__restrict__ ptr, const unsigned char *__restrict__ codes,
                                    int npat, const unsigned long long *__restrict__ hashes, const int *__restrict__ plen,
                                    const unsigned char *__restrict__ pcodes, unsigned char *__restrict__ rowpat,
                                    unsigned short *__restrict__ rowrel, int *__restrict__ bad)
{
    __shared__ unsigned long long hL[256];
    for (int i = threadIdx.x; i < npat; i += blockDim.x) hL[i] = hashes[i];
    __syncthreads();
    const int b = blockIdx.x, r0 = blk[b].x, k0 = blk[b].y, r1 = blk[b + 1].x;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += blockDim.x) {
        const int s = ptr[r], e = ptr[r + 1];
        const unsigned long long h = row_hash(codes, s, e);
        int lo = 0, hi = npat - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (hL[mid] < h) lo = mid + 1; else hi = mid; }
        bool ok = hL[lo] == h && plen[lo] == e - s && s - k0 < 65536;
        for (int k = s; ok && k < e; k++) ok = pcodes[lo * PAT_MAXLEN + (k - s)] == codes[k];
        if (!ok) atomicAdd(bad, 1);
        rowpat[r] = (unsigned char)lo;
        rowrel[r] = (unsigned short)(s - k0);
    }
}

// plan time: the set of (column - row) offsets, in a small open-addressing table; gives up beyond 255
constexpr int OFFSET_TABLE = 1024, OFFSET_EMPTY = -2147483647 - 1;
__global__ void csr_collect_offsets(int n, const int *__restrict__ ptr, const int *__restrict__ idx,
                                    int *__restrict__ table, int *__restrict__ count)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || count[0] > 255) return;
    for (int k = ptr[r]; k < ptr[r + 1]; k++) {
        const int off = idx[k] - r;
        unsigned h = ((unsigned)off * 2654435761u) >> 22;        // 10 bits
        for (int probe = 0; probe < OFFSET_TABLE; probe++) {
            const int v = table[h];
            if (v == off) break;
            if (v == OFFSET_EMPTY) {
                const int old = atomicCAS(&table[h], OFFSET_EMPTY, off);
                if (old == OFFSET_EMPTY) { atomicAdd(count, 1); break; }
                if (old == off) break;
            }
            h = (h + 1) & (OFFSET_TABLE - 1);
            if (count[0] > 255) return;
        }
    }
}

__global__ void csr_encode(int n, const int *__restrict__ ptr, const int *__restrict__ idx, const int *__restrict__ dict,
                           int ndict, unsigned char *__restrict__ codes)
{
    __shared__ int d[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) d[i] = dict[i];
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int k = ptr[r]; k < ptr[r + 1]; k++) {
        const int off = idx[k] - r;
        int lo = 0, hi = ndict - 1;                 // sorted ascending: the offset is there
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (d[mid] < off) lo = mid + 1; else hi = mid; }
        codes[k] = (unsigned char)lo;
    }
}

// value records: the values of each pattern's representative row ...
__global__ void csr_fetch_values(int npat, const int *__restrict__ rep, const int *__restrict__ ptr, const double *__restrict__ val,
                                 double *__restrict__ vrec, int stride = 8, int cap = 7)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npat) return;
    const int s = ptr[rep[p]], e = ptr[rep[p] + 1];
    for (int j = 0; j < stride; j++) vrec[stride * p + j] = (s + j < e && j < cap) ? val[s + j] : 0.0;
}
// ... and the check that EVERY row carries its pattern's values, bit for bit
__global__ void csr_check_values(int n, const int *__restrict__ ptr, const double *__restrict__ val,
                                 const unsigned char *__restrict__ rowpat, const double *__restrict__ vrec, int *__restrict__ bad,
                                 int stride = 8, int cap = 7)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || bad[0] != 0) return;
    const int s = ptr[r], e = ptr[r + 1], p = rowpat[r];
    bool ok = e - s <= cap;
    for (int k = s; ok && k < e; k++) ok = __double_as_longlong(val[k]) == __double_as_longlong(vrec[stride * p + (k - s)]);
    if (!ok) atomicAdd(bad, 1);
}

// Refinement: rows of one offset pattern that carry DIFFERENT values (a Dirichlet row stored with the interior row's sparsity, the
// explicit zeros of a DIA matrix) split the pattern.  One hash per row over (pattern, value bits), the distinct ones with their
// smallest row in the open-addressing table of the offset patterns' collector; gives up beyond 255.
__device__ __forceinline__ unsigned long long value_row_hash(int pat, const double *val, int s, int e)
{
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)(pat + 1);
    for (int k = s; k < e; k++) { h ^= (unsigned long long)__double_as_longlong(val[k]); h *= 1099511628211ull; h ^= h >> 29; }
    return h | 1ull;
}
__global__ void csr_collect_value_patterns(int n, const int *__restrict__ ptr, const double *__restrict__ val,
                                           const unsigned char *__restrict__ rowpat, unsigned long long *__restrict__ keys,
                                           int *__restrict__ rep, int *__restrict__ count)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || count[0] > 255) return;
    const unsigned long long h = value_row_hash(rowpat[r], val, ptr[r], ptr[r + 1]);
    unsigned slot = (unsigned)(h >> 40) & (PAT_SLOTS - 1);
    for (int probe = 0; probe < PAT_SLOTS; probe++) {
        unsigned long long v = keys[slot];
        if (v == 0ull) {
            v = atomicCAS(&keys[slot], 0ull, h);
            if (v == 0ull) { atomicAdd(count, 1); v = h; }
        }
        if (v == h) { if (r < rep[slot]) atomicMin(&rep[slot], r); return; }
        slot = (slot + 1) & (PAT_SLOTS - 1);
        if (count[0] > 255) return;
    }
}
// per refined pattern: the offset pattern of its representative row and that row's values
__global__ void csr_fetch_value_patterns(int npat, const int *__restrict__ rep, const int *__restrict__ ptr, const double *__restrict__ val,
                                         const unsigned char *__restrict__ rowpat, int *__restrict__ oldpat, double *__restrict__ vrec,
                                         int stride = 8, int cap = 7)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npat) return;
    const int s = ptr[rep[p]], e = ptr[rep[p] + 1];
    oldpat[p] = rowpat[rep[p]];
    for (int j = 0; j < stride; j++) vrec[stride * p + j] = (s + j < e && j < cap) ? val[s + j] : 0.0;
}
// every row -> its refined pattern (position of its hash in the sorted list; offset pattern and values verified)
__global__ void csr_encode_value_patterns(int n, const int *__restrict__ ptr, const double *__restrict__ val,
                                          const unsigned char *__restrict__ rowpat, int npat, const unsigned long long *__restrict__ hashes,
                                          const int *__restrict__ oldpat, const double *__restrict__ vrec,
                                          unsigned char *__restrict__ out, int *__restrict__ bad, int stride = 8, int cap = 7)
{
    __shared__ unsigned long long hL[256];
    for (int i = threadIdx.x; i < npat; i += blockDim.x) hL[i] = hashes[i];
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int s = ptr[r], e = ptr[r + 1], op = rowpat[r];
    const unsigned long long h = value_row_hash(op, val, s, e);
    int lo = 0, hi = npat - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hL[mid] < h) lo = mid + 1; else hi = mid; }
    bool ok = hL[lo] == h && oldpat[lo] == op && e - s <= cap;
    for (int k = s; ok && k < e; k++) ok = __double_as_longlong(val[k]) == __double_as_longlong(vrec[stride * lo + (k - s)]);
    if (!ok) atomicAdd(bad, 1);
    out[r] = (unsigned char)lo;
}

} // namespace

struct liship_csr_plan_s {
    int n;
    long long nnz;
    int nblocks;
    int geom;            // index into kGeom the split was built for
    int unroll;          // gather unroll U chosen from the mean row length
    int products;        // long rows on average: the products kernel (lanes own non-zeros) instead of row-gather
    int batch;           // its independent load pairs in flight per lane (2 or 4)
    v2i32 *blk;          // device, nblocks + 1 entries {row, ptr[row]}
    v2i32 *blk_host;     // host copy (row-range launches)
    unsigned char *codes; // device, one byte per non-zero (+ padding): position of (column - row) in dict; NULL = not coded
    int *dict;           // device, 256 sorted offsets (the tail repeats the last one)
    int ndict;
    unsigned short *lcol; // device, one 2 B position per non-zero into its row block's list of distinct columns; NULL = none
    int *dcol;           // device, the lists (each padded to a multiple of 4 entries)
    int *doff;           // device, nblocks + 1 offsets into dcol; an empty list = the block reads the 4 B indices
    long long ndcol;     // entries of dcol
    // Reordered form (liship_csr_plan_reorder): P A P^T in HBM -- the same entries in the same in-row order, rows and columns renumbered by landmark distances (csr_order.hpp) -- with a
    // block-local plan of its own.  The product gathers x into the new numbering, walks the renumbered rows and stores row r at y[r_perm[r]]: every row sum is the sum the
    // original row forms, term by term.
    liship_csr_plan_s *inner = nullptr;
    int *r_ptr = nullptr, *r_idx = nullptr, *r_perm = nullptr;      // device: row starts and columns of P A P^T; new position -> original row
    double *r_val = nullptr, *r_x = nullptr;                        // device: its values; x in the new numbering (n entries)
    int local_trial_failed = 0;                                     // a short-row plan tried block-local columns and its lists were too long (liship_csr_plan_localize_columns): a candidate for renumbering
    int ncols = 0;                                                  // > n: columns [n, ncols) are a rank's GHOST columns (liship_csr_plan_set_ghost_columns): the reordered form keeps them where they are
    int r_inner_end = 0;                                            // ... and puts the rows that read one behind all the others: rows [0, r_inner_end) of P A P^T touch no ghost column
    int *drun, *droff;   // device, or NULL: when every list is made of TRIPLES of consecutive columns (3 unknowns per node), the triples' first columns and nblocks + 1 offsets into them
    int ndpl;            // distinct columns per lane of spmv_csr_local_kernel: 2 (lists of <= 1024 columns) or 4
    int xcap;            // its x stage: the longest list rounded up to 1024 / 1536 / 2048 entries
    int xs_rows;         // rows per plane of a structured grid = the largest column offset of the row patterns (0: none): the XCD strips of the pattern kernels
    int first_term;      // row sums start at the first product instead of at +0.0 (split matrices)
    v4i32 *vrecw;        // device: WIDE value records for patterns of up to 32 entries (no ptab8): per pattern 144 B of byte offsets + length, 256 B of values; else NULL
    int *order;          // device, nblocks entries or NULL: launch order of the products kernel (blocks with a very long row first)
    v2i32 *tchunk = nullptr, *thead = nullptr;   // device, or NULL: tree mode's prepass (spmv_csr_tail_chunks_kernel) -- per chunk {row block, chunk of its tail}; per block with more
    double *tpart = nullptr;                     // than TAIL_FROM entries (+ a closing entry) {row block, its first chunk}; a partial sum per chunk
    int ntchunk = 0, nheavy = 0;
    unsigned char *rowpat; // device, one byte per row: its pattern (length + offset sequence); NULL = none
    unsigned short *rowrel; // device, 2 B per row: its first non-zero relative to its row block
    int *ptab;           // device: npat + 1 prefix entries, then the offsets of all patterns
    int ptab_len, npat;
    v4i32 *ptab8;        // device: when no pattern has more than 7 offsets, one 32 B record per pattern (7 offsets, length); else NULL
    v4i32 *prec_slot;    // device: with prec36, when ONE pattern carries most rows and its offsets are runs of equal length (box stencils): per pattern 32 slots
                         // into the wavefront's staged x (TeamRuns), length, "foreign" flag: spmv_csr_pattern_team_staged_kernel; else NULL
    TeamRuns tr;         // the runs of the dominant pattern (nruns = 0: none)
    double *wdrec;       // device: with vrecw, when one pattern carries most rows: per pattern WREC doubles (values in the dominant pattern's slots, mask | foreign << 32)
    v4i32 *wstage;       // device: 64 x 8 ints, the column offsets of a lane's slot pairs in the staging loads of spmv_csr_valuerecw_staged_kernel
    WideDom wd;          // the dominant wide pattern (len = 0: none)
    Box27 b27;           // the 27-point box stencil with constant coefficients (try_box27): the z-marching kernel's arguments (S = 0: none)
    Block2March b2;      // 2 x 2 block rows of a 7-point box grid (try_block2_march): the marching kernel's arguments (S = 0: none)
    BlockDom bd;         // block rows (liship_csr_plan_encode_block_rows): the dominant block row, one lane per block row (len = 0: none)
    unsigned long long *bdrec; // device: per pattern byte of a block row's FIRST row: mask over the dominant block row's entries | foreign << 32
    int *bstage;         // device: NL x 64 ints, the column offsets (from the wavefront's first row) of a lane's slot pairs in the staging loads
    v4i32 *prec36;       // device: when the longest pattern has 8..32 offsets, one 144 B record per pattern (32 byte offsets, length): spmv_csr_pattern_team_kernel; else NULL
    int prep[256];       // a row that carries each pattern
    v4i32 *vrec;         // device: with ptab8, when every row of a pattern carries the same values, 96 B per pattern (the 32 B record + 7 values); else NULL
    double *drec;        // device: with vrec, when one pattern dominates: per pattern 8 doubles {slots of the dominant pattern it has (mask; bit 7: not a
                         // subsequence), its values in those slots}; else NULL (spmv_csr_valuerec_dom_kernel)
    DomRec dom;          // the dominant pattern: byte offsets, values, pattern byte, slots
    int dom_xlen;        // 1 + the largest column the rows read (>= n; ghost columns in a multi-rank job): the marching kernel clamps its speculative loads into x[0, dom_xlen)
    int box_modes, box_pads;  // per neighbour kind (2 bits each: 0 -SO .. 6 +SO) what the box planes' rows do with a neighbour OUTSIDE the grid: 0 the slot is missing, 1 it is there with the
                         // dominant pattern's value (x real: a multi-rank job's ghost plane), 2 there with box_alt's value (DIA's explicit zeros); box_pads: a (row, +0.0) term per missing slot (ELL)
    double box_alt[7];   // by SLOT: the value of mode 2
    int box_z0, box_z1;  // planes [box_z0, box_z1) of the 7-point grid in which a slot is missing exactly where its neighbour lies outside the grid (dom_box_check): the marching kernel's BOX form
    int dom_simple;      // 1: every pattern is the dominant one's slots under a mask with the dominant one's VALUES (no foreign pattern, no padding terms): the marching kernel's short form
    int dom_lo, dom_hi;  // rows [dom_lo, dom_hi - 128] may start a wavefront that gathers x at the dominant offsets without leaving x[0, n)
};

static void build_dominant(liship_csr_plan_s *p, int npat, const int *rec32, const double *val8);

extern "C" int liship_spmv_csr_set_variant(int variant) { g_variant = variant; return 0; }
extern "C" int liship_spmv_csr_set_index_codes(int on) { g_index_codes = on ? 1 : 0; return 0; }
// On by default (round 6; it was opt-in before): the part of a row that does not fit the LDS stage (beyond ~2100 entries) is added by a workgroup-wide
// tree per pass instead of one strictly ordered chain.  Deterministic, but NOT bit-identical to the reference's left-to-right
// sum (within 1e-14 of the row's magnitude; north_star's bar for floating point is a tolerance, and the dots are trees already): a single row of 10^5 entries
// is a 10^5-long dependent add chain otherwise (5.5 ns a term: 7 % of the roofline on the heavy-tailed stress matrix).  0 restores the chain and with it the
// reference's bits for those rows (LIS_AMD_LONG_ROW_CHAIN=1; the reference-order reductions mode implies it).
extern "C" int liship_spmv_csr_set_long_row_tree(int on)
{
    const int v = on ? 1 : 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(d_long_row_tree), &v, sizeof(int)));
    g_long_row_tree_host = v;
    return 0;
}

// 1: the fused dots of the dominant-pattern product stay with the row blocks' partial sums (spmv_csr_valuerec_dom_dot4_kernel) -- the bits every other form of the
// product gives for its dots -- instead of a partial per tile of the plain product's kernel (faster; the same sums to rounding).  LIS_AMD_ROW_BLOCK_DOTS=1.
extern "C" int liship_spmv_csr_set_row_block_dots(int on) { g_row_block_dots = on ? 1 : 0; return 0; }
// the process-wide switches in force, for tests of the environment variables that set them: bit 0 team kernels, bit 1 row-block dots, bit 2 the long-row tree
extern "C" int liship_spmv_csr_switches(void) { return (g_team ? 1 : 0) | (g_row_block_dots ? 2 : 0) | (g_long_row_tree_host ? 4 : 0); }

// A/B switch of the uniform-length row sums (ordered_sum_rows): 0 keeps every wavefront on the skewed sums.  Same bits either way.
extern "C" int liship_spmv_csr_set_uniform_rows(int on)
{
    g_uniform_rows = on ? 1 : 0;
    return 0;
}

// the merge-path row split for the plan's geometry (device + host copy); replaces an existing one
static void free_tail(liship_csr_plan_s *p)
{
    if (p->tchunk) (void)hipFree(p->tchunk);
    if (p->thead) (void)hipFree(p->thead);
    if (p->tpart) (void)hipFree(p->tpart);
    p->tchunk = nullptr; p->thead = nullptr; p->tpart = nullptr; p->ntchunk = 0; p->nheavy = 0;
}
static int build_split(liship_csr_plan_s *p, const int *ptr, hipStream_t st)
{
    if (p->blk) { (void)hipFree(p->blk); p->blk = nullptr; }
    free(p->blk_host); p->blk_host = nullptr;
    const long long items = (long long)p->n + p->nnz;
    const int WORK = kGeom[p->geom].work;
    p->nblocks = (int)((items + WORK - 1) / WORK);
    if (p->nblocks <= 0) return 0;
    const size_t bytes = (size_t)(p->nblocks + 1) * sizeof(v2i32);
    hipError_t e = hipMalloc(&p->blk, bytes);
    if (e != hipSuccess) return (int)e;
    const int threads = 256, grid = (p->nblocks + 1 + threads - 1) / threads;
    csr_plan_kernel<<<grid, threads, 0, st>>>(p->n, ptr, p->nblocks, WORK, (g_variant & 0x1000000) ? 0 : 1, p->blk);
    e = hipGetLastError();
    p->blk_host = (v2i32 *)malloc(bytes);
    if (!p->blk_host) { (void)hipFree(p->blk); p->blk = nullptr; return LISHIP_ERR_ARG; }
    if (e == hipSuccess) e = hipMemcpyAsync(p->blk_host, p->blk, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(p->blk); p->blk = nullptr; free(p->blk_host); p->blk_host = nullptr; return (int)e; }
    free_tail(p);
    {                                                // blocks with more than TAIL_FROM entries: the chunks of their tails (tree mode's prepass)
        long long nch = 0; int nh = 0;
        for (int b = 0; b < p->nblocks; b++) {
            const int len = p->blk_host[b + 1].y - p->blk_host[b].y;
            if (len > TAIL_FROM) { nh++; nch += (len - TAIL_FROM + TAIL_CHUNK - 1) / TAIL_CHUNK; }
        }
        if (nh > 0) {
            v2i32 *hc = (v2i32 *)malloc(sizeof(v2i32) * (size_t)nch), *hh = (v2i32 *)malloc(sizeof(v2i32) * (size_t)(nh + 1));
            if (!hc || !hh) { free(hc); free(hh); return (int)hipErrorOutOfMemory; }
            int c = 0, h = 0;
            for (int b = 0; b < p->nblocks; b++) {
                const int len = p->blk_host[b + 1].y - p->blk_host[b].y;
                if (len <= TAIL_FROM) continue;
                hh[h].x = b; hh[h].y = c; h++;
                const int m = (len - TAIL_FROM + TAIL_CHUNK - 1) / TAIL_CHUNK;
                for (int j = 0; j < m; j++) { hc[c].x = b; hc[c].y = j; c++; }
            }
            hh[nh].x = p->nblocks - 1; hh[nh].y = c;
            e = hipMalloc(&p->tchunk, sizeof(v2i32) * (size_t)nch);
            if (e == hipSuccess) e = hipMalloc(&p->thead, sizeof(v2i32) * (size_t)(nh + 1));
            if (e == hipSuccess) e = hipMalloc(&p->tpart, sizeof(double) * (size_t)nch);
            if (e == hipSuccess) e = hipMemcpy(p->tchunk, hc, sizeof(v2i32) * (size_t)nch, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->thead, hh, sizeof(v2i32) * (size_t)(nh + 1), hipMemcpyHostToDevice);
            free(hc); free(hh);
            if (e != hipSuccess) { free_tail(p); return (int)e; }      // (tree mode's kernels count on the prepass: no plan without it)
            p->ntchunk = (int)nch; p->nheavy = nh;
        }
    }
    if (p->order) { (void)hipFree(p->order); p->order = nullptr; }
    if (p->products && p->nblocks > 1) {             // blocks whose last row overflows the stage by far: launched first (products kernel)
        const int heavy_from = 4 * (WORK + SLACK);
        int nheavy = 0;
        for (int b = 0; b < p->nblocks; b++) nheavy += (p->blk_host[b + 1].y - p->blk_host[b].y > heavy_from);
        if (nheavy > 0 && nheavy < p->nblocks) {
            int *ord = (int *)malloc(sizeof(int) * (size_t)p->nblocks);
            if (ord) {
                int at = 0;
                for (int b = 0; b < p->nblocks; b++) if (p->blk_host[b + 1].y - p->blk_host[b].y > heavy_from) ord[at++] = b;
                for (int i = 1; i < nheavy; i++) {       // the longest first
                    const int v = ord[i]; int j = i - 1;
                    auto len = [&](int b) { return p->blk_host[b + 1].y - p->blk_host[b].y; };
                    while (j >= 0 && len(ord[j]) < len(v)) { ord[j + 1] = ord[j]; j--; }
                    ord[j + 1] = v;
                }
                for (int b = 0; b < p->nblocks; b++) if (!(p->blk_host[b + 1].y - p->blk_host[b].y > heavy_from)) ord[at++] = b;
                if (hipMalloc(&p->order, sizeof(int) * (size_t)p->nblocks) == hipSuccess) {
                    if (hipMemcpy(p->order, ord, sizeof(int) * (size_t)p->nblocks, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->order); p->order = nullptr; }
                } else { p->order = nullptr; (void)hipGetLastError(); }
                free(ord);
            }
        }
    }
    return 0;
}

extern "C" int liship_csr_plan_create(liship_csr_plan_t *out, int n, const int *ptr, void *stream)
{
    if (!out || n < 0 || (n > 0 && !ptr)) return LISHIP_ERR_ARG;
    hipStream_t st = as_stream(stream);
    int nnz = 0;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(&nnz, ptr + n, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    liship_csr_plan_s *p = new liship_csr_plan_s();
    p->n = n;
    p->nnz = nnz;
    p->geom = (g_variant >> 4) & 15;
    if (p->geom >= kNumGeom || p->geom == 2 || p->geom == 3 || p->geom == 4 || p->geom == 6) { delete p; return LISHIP_ERR_ARG; }
    const double mean_len = n > 0 ? (double)nnz / n : 0.0;
    p->unroll = mean_len <= 4.0 ? 4 : (mean_len <= 7.0 ? 7 : 8);
    // lane-per-row keeps 176 lanes busy on 7-entry rows but only 17 on 80-entry rows: from 22 entries per row on,
    // lanes own non-zeros instead (measured crossover with U = 8, tools/rowlen_sweep.py: the row-gather kernel wins up to
    // 20 entries per row, ties at 22-24, loses from 27 on).  With coded indices (liship_csr_plan_encode_indices) the
    // row-gather kernel stays ahead up to 29 entries per row and the plan switches back to it.
    p->products = (g_variant == 0 && mean_len >= 22.0) ? 1 : 0;
    if (p->products) p->geom = 1;
    p->batch = mean_len >= 24.0 ? 2 : 4;
    p->blk = nullptr;
    p->blk_host = nullptr;
    p->codes = nullptr; p->dict = nullptr; p->ndict = 0;
    p->lcol = nullptr; p->dcol = nullptr; p->doff = nullptr; p->drun = nullptr; p->droff = nullptr; p->ndcol = 0; p->ndpl = 2;
    p->first_term = 0;
    p->rowpat = nullptr; p->rowrel = nullptr; p->ptab = nullptr; p->ptab_len = 0; p->npat = 0; p->ptab8 = nullptr; p->prec36 = nullptr; p->prec_slot = nullptr; p->tr.nruns = 0; p->wdrec = nullptr; p->wstage = nullptr; p->wd.len = 0; p->b27.S = 0; p->b2.S = 0; p->bd.len = 0; p->bdrec = nullptr; p->bstage = nullptr; p->vrec = nullptr; p->vrecw = nullptr; p->order = nullptr;
    p->drec = nullptr; p->dom_lo = p->dom_hi = 0; p->dom_simple = 0; p->box_z0 = p->box_z1 = 0; p->dom_xlen = 0; p->box_modes = p->box_pads = 0;
    const int rc = build_split(p, ptr, st);
    if (rc) { delete p; return rc; }
    *out = p;
    return 0;
}

extern "C" int liship_csr_plan_destroy(liship_csr_plan_t p)
{
    if (!p) return 0;
    int rc = 0;
    if (p->blk) rc = (int)hipFree(p->blk);
    if (p->codes) (void)hipFree(p->codes);
    if (p->dict) (void)hipFree(p->dict);
    if (p->rowpat) (void)hipFree(p->rowpat);
    if (p->rowrel) (void)hipFree(p->rowrel);
    if (p->ptab) (void)hipFree(p->ptab);
    if (p->ptab8) (void)hipFree(p->ptab8);
    if (p->prec36) (void)hipFree(p->prec36);
    if (p->prec_slot) (void)hipFree(p->prec_slot);
    if (p->wdrec) (void)hipFree(p->wdrec);
    if (p->wstage) (void)hipFree(p->wstage);
    if (p->bdrec) (void)hipFree(p->bdrec);
    if (p->bstage) (void)hipFree(p->bstage);
    if (p->vrec) (void)hipFree(p->vrec);
    if (p->drec) (void)hipFree(p->drec);
    if (p->order) (void)hipFree(p->order);
    free_tail(p);
    if (p->vrecw) (void)hipFree(p->vrecw);
    if (p->lcol) (void)hipFree(p->lcol);
    if (p->dcol) (void)hipFree(p->dcol);
    if (p->drun) (void)hipFree(p->drun);
    if (p->droff) (void)hipFree(p->droff);
    if (p->doff) (void)hipFree(p->doff);
    if (p->inner) (void)liship_csr_plan_destroy(p->inner);
    if (p->r_ptr) (void)hipFree(p->r_ptr);
    if (p->r_idx) (void)hipFree(p->r_idx);
    if (p->r_perm) (void)hipFree(p->r_perm);
    if (p->r_val) (void)hipFree(p->r_val);
    if (p->r_x) (void)hipFree(p->r_x);
    free(p->blk_host);
    delete p;
    return rc;
}

// the rows of this matrix are chains that START with their first product (the split form D x + L x + U x of the reference,
// lis_matvec_csr.c:64-89) instead of being added to 0.0: same kernels, the running sum starts at -0.0
extern "C" int liship_csr_plan_set_first_term_initialises(liship_csr_plan_t p, int on)
{
    if (!p) return LISHIP_ERR_ARG;
    p->first_term = on ? 1 : 0;
    if (p->inner) p->inner->first_term = p->first_term;
    return 0;
}

extern "C" int liship_csr_plan_info(liship_csr_plan_t p, int *n, long long *nnz, int *nblocks)
{
    if (!p) return LISHIP_ERR_ARG;
    if (n) *n = p->n;
    if (nnz) *nnz = p->nnz;
    if (nblocks) *nblocks = p->nblocks;
    return 0;
}

// One byte per column index where the matrix allows it (see spmv_csr_coded_kernel): at most 255 distinct
// (column - row) offsets, short rows on average (the row-gather kernel), 16 B aligned arrays.  Setup-time: two passes
// over ptr / index.  Not an error when the matrix does not qualify -- the plan then keeps using the index array.
extern "C" int liship_csr_plan_encode_indices(liship_csr_plan_t p, const int *ptr, const int *idx, void *stream)
{
    if (!p || (p->n > 0 && (!ptr || !idx))) return LISHIP_ERR_ARG;
    const double mean_len = (double)p->nnz / (p->n > 0 ? p->n : 1);
    if (p->codes || p->n == 0 || p->nnz == 0 || !aligned16(idx) || (g_variant != 0 && p->products)) return 0;
    if (p->products && mean_len >= 30.0) return 0;   // long rows: the products kernel, which reads the 4 B indices
    hipStream_t st = as_stream(stream);
    int *table = nullptr;                            // OFFSET_TABLE slots + the counter
    HIP_TRY(hipMalloc(&table, sizeof(int) * (OFFSET_TABLE + 1)));
    int host[OFFSET_TABLE + 1];
    for (int i = 0; i < OFFSET_TABLE; i++) host[i] = OFFSET_EMPTY;
    host[OFFSET_TABLE] = 0;
    hipError_t e = hipMemcpyAsync(table, host, sizeof(host), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        csr_collect_offsets<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, idx, table, table + OFFSET_TABLE);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, table, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(table);
    if (e != hipSuccess) return (int)e;
    if (host[OFFSET_TABLE] > 255) return 0;          // too many diagonals: stay with 4 B indices
    int dict[256], nd = 0;
    for (int i = 0; i < OFFSET_TABLE; i++) if (host[i] != OFFSET_EMPTY && nd < 256) dict[nd++] = host[i];
    if (nd == 0 || nd > 255) return 0;
    for (int i = 1; i < nd; i++) { const int v = dict[i]; int j = i - 1; while (j >= 0 && dict[j] > v) { dict[j + 1] = dict[j]; j--; } dict[j + 1] = v; }
    for (int i = nd; i < 256; i++) dict[i] = dict[nd - 1];
    const size_t cbytes = ((size_t)p->nnz + 15) / 16 * 16 + 16 * WAVE;       // whole 16 B pieces, one wave slice of slack
    e = hipMalloc(&p->dict, sizeof(dict));
    if (e == hipSuccess) e = hipMalloc(&p->codes, cbytes);
    if (e == hipSuccess) e = hipMemsetAsync(p->codes, 0, cbytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(p->dict, dict, sizeof(dict), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        csr_encode<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, idx, p->dict, nd, p->codes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        if (p->codes) (void)hipFree(p->codes);
        if (p->dict) (void)hipFree(p->dict);
        p->codes = nullptr; p->dict = nullptr;
        return (int)e;
    }
    p->ndict = nd;
    // 9 B per item instead of 12: the coded kernel runs best on 256-lane workgroups of 2048 items (tools/coded_sweep.py),
    // and up to 29 entries per row it beats the products kernel the plan had chosen for 22+
    if (g_variant == 0 && (p->geom == 0 || p->products)) { p->geom = 1; p->products = 0; return build_split(p, ptr, st); }
    return 0;
}
// number of dictionary entries when the plan's indices are coded, 0 otherwise
extern "C" int liship_csr_plan_coded(liship_csr_plan_t p) { return (p && p->codes) ? p->ndict : 0; }

// Row patterns on top of the column codes (see spmv_csr_pattern_kernel): setup-time, optional, never an error when the matrix
// does not qualify (not coded, a row longer than 64, more than 255 patterns, more than 1024 offsets over all patterns).
// Must follow liship_csr_plan_encode_indices, whose final row split it encodes the row starts against.
// the 144 B records of spmv_csr_pattern_team_kernel from a host copy of the pattern table T (NP + 1 prefix entries, then element offsets):
// per pattern 32 column offsets IN BYTES (the tail repeats the last: always a column of the row) and the length.  Kept when the longest
// pattern has 8..32 offsets and none is empty; rebuilt whenever the plan's pattern bytes are renumbered.  Never an error.
static void build_team_records(liship_csr_plan_s *p, const int *T, int NP)
{
    if (p->prec36) { (void)hipFree(p->prec36); p->prec36 = nullptr; }
    int maxlen = 0, minlen = 1 << 30, maxoff = 0;
    for (int i = 0; i < NP; i++) { const int l = T[i + 1] - T[i]; if (l > maxlen) maxlen = l; if (l < minlen) minlen = l; }
    for (int t = 0; t < T[NP]; t++) if (T[NP + 1 + t] > maxoff) maxoff = T[NP + 1 + t];
    if (NP <= 0 || maxlen <= 7 || maxlen > TEAM_MAXLEN || minlen < 1 || (long long)p->n + maxoff >= (1ll << 28)) return;
    int *rec = (int *)calloc((size_t)NP * 4 * TEAM_REC, sizeof(int));
    if (!rec) return;
    for (int i = 0; i < NP; i++) {
        const int l = T[i + 1] - T[i];
        for (int j = 0; j < TEAM_MAXLEN; j++) rec[4 * TEAM_REC * i + j] = 8 * T[NP + 1 + T[i] + (j < l ? j : l - 1)];
        rec[4 * TEAM_REC * i + TEAM_MAXLEN] = l;
    }
    if (hipMalloc(&p->prec36, sizeof(int) * 4 * TEAM_REC * (size_t)NP) == hipSuccess) {
        if (hipMemcpy(p->prec36, rec, sizeof(int) * 4 * TEAM_REC * (size_t)NP, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->prec36); p->prec36 = nullptr; }
    } else p->prec36 = nullptr;
    free(rec);
}

__global__ void rowpat_histogram(int n, const unsigned char *__restrict__ rowpat, unsigned long long *__restrict__ count);

static void build_wide_dominant(liship_csr_plan_s *p, const int *T, int NP, const double *vals, const int *ptr, hipStream_t st);

// largest column index of a coded matrix (the staged x of the team kernel is read speculatively: its addresses are clamped to the array)
__global__ void csr_max_column(int n, const int *__restrict__ ptr, const unsigned char *__restrict__ codes, const int *__restrict__ dict, int *__restrict__ out)
{
    int m = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        for (int k = ptr[r]; k < ptr[r + 1]; k++) m = max(m, r + dict[codes[k]]);
    atomicMax(out, m);
}

// The staged-x form of the four-lanes-per-row kernel (spmv_csr_pattern_team_staged_kernel).  When ONE pattern carries most rows, its offsets,
// sorted, fall into at most 16 runs of consecutive columns (9 runs of 3 for the 27-point stencil; 1, 3, 1, 3, 3, 3, 1, 3, 1 for the 19-point one), and the
// x values that 16 neighbouring rows need from a run of m are 15 + m consecutive doubles: a wavefront stages them -- the runs' "slots", at most 254 --
// with ceil(slots / 128) coalesced 16 B loads instead of one gather per entry, and a row's entry reads slot base[run] + (offset - run start) + its row.
// Every pattern whose offsets all lie inside the dominant one's runs is served from the same slots (the boundary rows of a stencil);
// the others are flagged foreign and gather for themselves.  Called when the plan's pattern bytes are final; never an error.
static void build_team_runs(liship_csr_plan_s *p, const int *ptr, hipStream_t st)
{
    if (p->prec_slot) { (void)hipFree(p->prec_slot); p->prec_slot = nullptr; }
    p->tr.nruns = 0;
    const int NP = p->npat;
    if (!p->prec36 || !p->rowpat || !p->ptab || NP <= 0 || NP > 255 || p->n < 4 * WAVE || !p->codes || !p->dict) return;
    int *T = (int *)malloc(sizeof(int) * (size_t)p->ptab_len);
    unsigned long long *d_count = nullptr, count[256];
    int *d_max = nullptr, maxcol = 0;
    bool ok = T && hipMalloc(&d_count, sizeof(count)) == hipSuccess && hipMalloc(&d_max, sizeof(int)) == hipSuccess;
    ok = ok && hipMemsetAsync(d_count, 0, sizeof(count), st) == hipSuccess && hipMemsetAsync(d_max, 0, sizeof(int), st) == hipSuccess;
    if (ok) { rowpat_histogram<<<1024, 256, 0, st>>>(p->n, p->rowpat, d_count); ok = hipGetLastError() == hipSuccess; }
    if (ok) { csr_max_column<<<2048, 256, 0, st>>>(p->n, ptr, p->codes, p->dict, d_max); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(count, d_count, sizeof(count), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(&maxcol, d_max, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(T, p->ptab, sizeof(int) * (size_t)p->ptab_len, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (d_count) (void)hipFree(d_count);
    if (d_max) (void)hipFree(d_max);
    if (ok) {
        int dom = 0, maxlen = 0;
        for (int i = 1; i < NP; i++) if (count[i] > count[dom]) dom = i;
        for (int i = 0; i < NP; i++) maxlen = max(maxlen, T[i + 1] - T[i]);
        const int l = T[dom + 1] - T[dom];
        int offs[TEAM_MAXLEN], start[16], mlen[16], base[17], nruns = 0;
        ok = count[dom] * 2 >= (unsigned long long)p->n && l >= 1 && l <= TEAM_MAXLEN;
        if (ok) {
            for (int j = 0; j < l; j++) offs[j] = T[NP + 1 + T[dom] + j];
            for (int a = 1; a < l; a++) { const int v = offs[a]; int b = a - 1; while (b >= 0 && offs[b] > v) { offs[b + 1] = offs[b]; b--; } offs[b + 1] = v; }
            for (int j = 0; j < l && ok; ) {                 // runs of consecutive offsets
                int e = j + 1;
                while (e < l && offs[e] == offs[e - 1] + 1) e++;
                if (e < l && offs[e] == offs[e - 1]) ok = false;     // (a repeated offset: not this kernel)
                if (nruns == 16) ok = false; else { start[nruns] = offs[j]; mlen[nruns++] = e - j; }
                j = e;
            }
        }
        // 16 neighbouring rows need 15 + m consecutive columns of a run of m; widths are rounded up to even (the staging lanes take pairs of slots)
        base[0] = 0;
        for (int a = 0; a < nruns; a++) base[a + 1] = base[a] + ((15 + mlen[a] + 1) & ~1);
        const int slots = base[nruns];
        ok = ok && nruns >= 1 && slots <= 254 && maxcol >= 1 && l > nruns;       // (slots are bytes in the records; no run longer than one column: nothing to share)
        // one 16 B record per (pattern, team lane t): the eight slots of entries 8t .. 8t+7 as bytes (the tail repeats a valid slot), the row's
        // length, the foreign flag -- a single load per lane
        unsigned char *rec8 = ok ? (unsigned char *)calloc((size_t)NP * 4 * 16, 1) : nullptr;
        if (ok && rec8) {
            for (int i = 0; i < NP; i++) {
                const int li = T[i + 1] - T[i];
                int foreign = 0, slot[TEAM_MAXLEN];
                for (int j = 0; j < TEAM_MAXLEN; j++) slot[j] = 0;
                for (int j = 0; j < li; j++) {
                    const int o = T[NP + 1 + T[i] + j];
                    int q = -1;
                    for (int a = 0; a < nruns; a++) if (o >= start[a] && o < start[a] + mlen[a]) q = a;
                    if (q < 0) { foreign = 1; break; }
                    slot[j] = base[q] + (o - start[q]);
                }
                for (int t = 0; t < 4; t++) {
                    unsigned char *r16 = rec8 + ((size_t)i * 4 + t) * 16;
                    for (int u = 0; u < 8; u++) r16[u] = (unsigned char)(foreign ? 0 : slot[8 * t + u]);
                    r16[8] = (unsigned char)li; r16[9] = (unsigned char)foreign;
                }
            }
            if (hipMalloc(&p->prec_slot, (size_t)NP * 4 * 16) == hipSuccess &&
                hipMemcpy(p->prec_slot, rec8, (size_t)NP * 4 * 16, hipMemcpyHostToDevice) == hipSuccess) {
                p->tr.nruns = nruns; p->tr.slots = slots; p->tr.maxcol = maxcol; p->tr.maxlen = maxlen;
                for (int a = 0; a < 16; a++) { p->tr.start[a] = a < nruns ? start[a] : 0; p->tr.base[a] = a < nruns ? base[a] : (1 << 20); }
            } else if (p->prec_slot) { (void)hipFree(p->prec_slot); p->prec_slot = nullptr; }
        }
        free(rec8);
    }
    free(T);
}

extern "C" int liship_csr_plan_encode_row_patterns(liship_csr_plan_t p, const int *ptr, void *stream)
{
    if (!p || (p->n > 0 && !ptr)) return LISHIP_ERR_ARG;
    if (!p->codes || p->rowpat || p->products || p->nblocks <= 0 || g_variant != 0) return 0;
    hipStream_t st = as_stream(stream);
    struct Table { unsigned long long keys[PAT_SLOTS]; int rep[PAT_SLOTS]; int count; };
    Table *host = (Table *)malloc(sizeof(Table));
    Table *dev = nullptr;
    if (!host) return LISHIP_ERR_ARG;
    hipError_t e = hipMalloc(&dev, sizeof(Table));
    if (e != hipSuccess) { free(host); return (int)e; }
    memset(host, 0, sizeof(Table));
    for (int i = 0; i < PAT_SLOTS; i++) host->rep[i] = 0x7fffffff;
    e = hipMemcpyAsync(dev, host, sizeof(Table), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        csr_collect_patterns<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, p->codes, dev->keys, dev->rep, &dev->count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(Table), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) { free(host); return (int)e; }
    if (host->count > 255 || host->count <= 0) { free(host); return 0; }
    // the patterns, sorted by hash (what csr_encode_patterns searches)
    int npat = 0, order[256];
    for (int i = 0; i < PAT_SLOTS; i++) if (host->keys[i] != 0ull && npat < 256) order[npat++] = i;
    for (int i = 1; i < npat; i++) { const int v = order[i]; int j = i - 1; while (j >= 0 && host->keys[order[j]] > host->keys[v]) { order[j + 1] = order[j]; j--; } order[j + 1] = v; }
    unsigned long long hashes[256]; int reps[256], plen[256];
    for (int i = 0; i < npat; i++) { hashes[i] = host->keys[order[i]]; reps[i] = host->rep[order[i]]; }
    free(host);
    unsigned long long *d_hash = nullptr; int *d_rep = nullptr, *d_len = nullptr, *d_bad = nullptr; unsigned char *d_pc = nullptr;
    unsigned char *pcodes = (unsigned char *)calloc((size_t)npat * PAT_MAXLEN, 1);
    int rc = 0, bad = 0, dict[256];
#define PT(expr) do { if (rc == 0) { hipError_t e__ = (expr); if (e__ != hipSuccess) rc = (int)e__; } } while (0)
    PT(hipMalloc(&d_hash, sizeof(hashes))); PT(hipMalloc(&d_rep, sizeof(reps))); PT(hipMalloc(&d_len, sizeof(plen)));
    PT(hipMalloc(&d_bad, sizeof(int))); PT(hipMalloc(&d_pc, (size_t)npat * PAT_MAXLEN));
    PT(hipMemcpyAsync(d_hash, hashes, sizeof(unsigned long long) * npat, hipMemcpyHostToDevice, st));
    PT(hipMemcpyAsync(d_rep, reps, sizeof(int) * npat, hipMemcpyHostToDevice, st));
    PT(hipMemsetAsync(d_bad, 0, sizeof(int), st));
    PT(hipMemsetAsync(d_pc, 0, (size_t)npat * PAT_MAXLEN, st));
    if (rc == 0) { csr_fetch_patterns<<<1, 256, 0, st>>>(npat, d_rep, ptr, p->codes, d_len, d_pc); PT(hipGetLastError()); }
    PT(hipMemcpyAsync(plen, d_len, sizeof(int) * npat, hipMemcpyDeviceToHost, st));
    PT(hipMemcpyAsync(pcodes, d_pc, (size_t)npat * PAT_MAXLEN, hipMemcpyDeviceToHost, st));
    PT(hipMemcpyAsync(dict, p->dict, sizeof(dict), hipMemcpyDeviceToHost, st));
    PT(hipStreamSynchronize(st));
    int total = 0;
    for (int i = 0; i < npat && rc == 0; i++) total += plen[i];
    // the one-lane-per-row pattern kernel keeps the table in LDS (PAT_TABLE ints); patterns of 8..32 offsets run on the team kernels, which read 16 B /
    // 144 B records instead: for them a larger table is accepted (125 patterns of up to 13 for the fourth-order star in 3-D), and the general kernel is
    // never launched on such a plan (launch_geom / launch_rowgather_dot fall through to the coded kernel when the A/B switches turn the teams off)
    int maxl = 0, minl = 1 << 30;
    for (int i = 0; i < npat && rc == 0; i++) { maxl = max(maxl, plen[i]); minl = min(minl, plen[i]); }
    const bool teams = maxl > 7 && maxl <= TEAM_MAXLEN && minl >= 1;
    const bool fits = rc == 0 && ((total <= 1024 && npat + 1 + total <= PAT_TABLE) || (teams && total <= 4096));
    int *tab = nullptr;
    if (fits) {
        tab = (int *)malloc(sizeof(int) * (size_t)(npat + 1 + total));
        int at = 0;
        for (int i = 0; i < npat; i++) { tab[i] = at; for (int j = 0; j < plen[i]; j++) tab[npat + 1 + at + j] = dict[pcodes[i * PAT_MAXLEN + j]]; at += plen[i]; }
        tab[npat] = at;
        PT(hipMalloc(&p->ptab, sizeof(int) * (size_t)(npat + 1 + total)));
        PT(hipMalloc(&p->rowpat, (size_t)p->n + 64));
        PT(hipMalloc(&p->rowrel, sizeof(unsigned short) * ((size_t)p->n + 64)));
        PT(hipMemcpyAsync(p->ptab, tab, sizeof(int) * (size_t)(npat + 1 + total), hipMemcpyHostToDevice, st));
        int maxlen = 0;
        for (int i = 0; i < npat; i++) if (plen[i] > maxlen) maxlen = plen[i];
        int minlen = 1 << 30;
        for (int i = 0; i < npat; i++) if (plen[i] < minlen) minlen = plen[i];
        int maxoff = 0;                             // the kernel addresses x by 32-bit byte offsets: row + offset < 2^29
        for (int t = 0; t < total; t++) if (tab[npat + 1 + t] > maxoff) maxoff = tab[npat + 1 + t];
        if (maxlen <= 7 && minlen >= 1 && npat <= PAT7_MAX && (long long)p->n + maxoff < (1ll << 29)) {
            // one 32 B record per pattern: the 7 column offsets IN BYTES (the tail repeats the last: always a column of the row), the length
            int rec[PAT7_MAX * 8];
            for (int i = 0; i < npat; i++) {
                for (int j = 0; j < 7; j++) rec[8 * i + j] = 8 * tab[npat + 1 + tab[i] + (j < plen[i] ? j : plen[i] - 1)];
                rec[8 * i + 7] = plen[i];
            }
            PT(hipMalloc(&p->ptab8, sizeof(int) * 8 * (size_t)npat));
            PT(hipMemcpyAsync(p->ptab8, rec, sizeof(int) * 8 * (size_t)npat, hipMemcpyHostToDevice, st));
            for (int i = 0; i < npat; i++) p->prep[i] = reps[i];
        }
        // a structured grid: the plane the XCD strips are cut from (xcd_strip_unit).  The positive offsets of the longest pattern fall into clusters -- the line's
        // neighbours, the neighbouring lines, the neighbouring planes --; the plane is the centre of the cluster above the LARGEST gap (7-point: +SO itself; 27-point:
        // SO - S - 1 .. SO + S + 1 -> SO; 9-point in 2-D: S - 1 .. S + 1 -> S).  A pattern with a single positive cluster keeps the largest offset.
        {
            int lp = 0;
            for (int i = 1; i < npat; i++) if (plen[i] > plen[lp]) lp = i;
            int pos[PAT_MAXLEN], np_ = 0;
            for (int j = 0; j < plen[lp]; j++) { const int o = tab[npat + 1 + tab[lp] + j]; if (o > 0 && np_ < PAT_MAXLEN) pos[np_++] = o; }
            for (int i = 1; i < np_; i++) { const int v = pos[i]; int j = i - 1; while (j >= 0 && pos[j] > v) { pos[j + 1] = pos[j]; j--; } pos[j + 1] = v; }
            int plane = maxoff;
            if (np_ >= 2) {
                int g = 1;
                for (int i = 2; i < np_; i++) if (pos[i] - pos[i - 1] > pos[g] - pos[g - 1]) g = i;
                if (pos[g] - pos[g - 1] > 2) plane = (pos[g] + pos[np_ - 1]) / 2;
            }
            p->xs_rows = plane;
        }
        if (rc == 0) build_team_records(p, tab, npat);
        if (rc == 0) {
            csr_encode_patterns<<<p->nblocks, 256, 0, st>>>(p->blk, ptr, p->codes, npat, d_hash, d_len, d_pc, p->rowpat, p->rowrel, d_bad);
            PT(hipGetLastError());
        }
        PT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
        PT(hipStreamSynchronize(st));
    }
#undef PT
    (void)hipFree(d_hash); (void)hipFree(d_rep); (void)hipFree(d_len); (void)hipFree(d_bad); (void)hipFree(d_pc);
    free(pcodes); free(tab);
    if (rc != 0 || !fits || bad != 0) {             // not this matrix (or two patterns with one hash): the codes serve
        if (p->ptab) (void)hipFree(p->ptab);
        if (p->rowpat) (void)hipFree(p->rowpat);
        if (p->rowrel) (void)hipFree(p->rowrel);
        if (p->ptab8) (void)hipFree(p->ptab8);
        if (p->prec36) (void)hipFree(p->prec36);
        if (p->prec_slot) (void)hipFree(p->prec_slot);
        p->prec_slot = nullptr; p->tr.nruns = 0;
        p->ptab = nullptr; p->rowpat = nullptr; p->rowrel = nullptr; p->ptab8 = nullptr; p->prec36 = nullptr;
        return rc;
    }
    p->npat = npat; p->ptab_len = npat + 1 + total;
    for (int i = 0; i < npat; i++) p->prep[i] = reps[i];
    build_team_runs(p, ptr, st);                    // (patterns of 8..32 offsets: the staged-x form of the four-lanes-per-row kernel, when one pattern dominates)
    if (p->prec36) {                                // ... and the lane-per-row form with the dominant pattern's slots in scalar registers
        int *Th = (int *)malloc(sizeof(int) * (size_t)p->ptab_len);
        if (Th && hipMemcpy(Th, p->ptab, sizeof(int) * (size_t)p->ptab_len, hipMemcpyDeviceToHost) == hipSuccess) build_wide_dominant(p, Th, p->npat, nullptr, ptr, st);
        free(Th);
    }
    if (p->ptab8) {                                 // the dominant pattern, offsets only (values join with the value records)
        int rec8[PAT7_MAX * 8];
        if (hipMemcpy(rec8, p->ptab8, sizeof(int) * 8 * (size_t)npat, hipMemcpyDeviceToHost) == hipSuccess) build_dominant(p, npat, rec8, nullptr);
    }
    return 0;
}
// number of row patterns when the plan keeps one byte per row, 0 otherwise
extern "C" int liship_csr_plan_row_patterns(liship_csr_plan_t p) { return (p && p->rowpat) ? p->npat : 0; }
// 1 when every pattern has 1..7 offsets and the plan also keeps them as 32 B records (spmv_csr_pattern7_kernel), else 0
// 1 when the plan keeps the 144 B records of spmv_csr_pattern_team_kernel (longest pattern 8..32 offsets)
extern "C" int liship_csr_plan_team_records(liship_csr_plan_t p) { return (p && p->rowpat && p->prec36) ? 1 : 0; }
// 2 when the plan also keeps the dominant pattern's runs and slot records (the staged-x form), 1: records only, 0: none
// 1 when a plan with wide value records also keeps the dominant pattern for the staged-x kernel (spmv_csr_valuerecw_staged_kernel)
// 1 when the whole-matrix product of a plan with wide value records marches (the 27-point box stencil: spmv_csr_box27_march_kernel) under the switches in force
extern "C" int liship_csr_plan_box27(liship_csr_plan_t p);
extern "C" int liship_csr_plan_wide_dominant(liship_csr_plan_t p) { return (p && p->vrecw && p->wdrec && p->wd.len > 0) ? 1 : 0; }
extern "C" int liship_csr_plan_team_form(liship_csr_plan_t p) { return (p && p->rowpat && p->prec36) ? ((p->prec_slot && p->tr.nruns > 0) ? 2 : 1) : 0; }
extern "C" int liship_csr_plan_pattern_records(liship_csr_plan_t p) { return (p && p->rowpat && p->ptab8) ? 1 : 0; }
extern "C" int liship_spmv_csr_set_row_patterns(int on) { g_row_patterns = on ? 1 : 0; return 0; }

// plan time: how many rows carry each pattern byte
__global__ void rowpat_histogram(int n, const unsigned char *__restrict__ rowpat, unsigned long long *__restrict__ count)
{
    __shared__ unsigned int h[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) h[t] = 0;
    __syncthreads();
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) atomicAdd(&h[rowpat[r]], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < 256; t += blockDim.x) if (h[t]) atomicAdd(&count[t], (unsigned long long)h[t]);
}

// the 7-point stencil?  S, SO: the strides of the middle and the outer pair of offsets; perm: 3 bits per slot, which neighbour it is (0: -SO, 1: -S, 2: -1, 3: 0, 4: +1, 5: +S, 6: +SO)
static bool dom_seven_point(const DomRec &D, int &S, int &SO, int &perm)
{
    if (D.mask != 0x7f) return false;
    S = 0; SO = 0;
    for (int u = 0; u < 7; u++) {                              // (dom_stride / dom_stride_outer, defined further down: the smallest and the largest offset beyond +-1 present on both sides)
        const int e = D.off[u] / 8;
        bool both = false;
        for (int v = 0; v < 7; v++) both = both || D.off[v] == -8 * e;
        if (e > 1 && both) { if (S == 0 || e < S) S = e; if (e > SO) SO = e; }
    }
    if (S <= 1 || SO <= S) return false;
    perm = 0;
    int seen = 0;
    for (int u = 0; u < 7; u++) {
        const int e = D.off[u] / 8;
        const int k = e == -SO ? 0 : e == -S ? 1 : e == -1 ? 2 : e == 0 ? 3 : e == 1 ? 4 : e == S ? 5 : e == SO ? 6 : -1;
        if (k < 0 || D.off[u] % 8 != 0) return false;
        perm |= k << (3 * u); seen |= 1 << k;
    }
    return seen == 0x7f;
}
// plan time: the largest column any row reads (by its pattern's largest offset): the x entries a speculative load may touch are [0, that] -- in a multi-rank job
// beyond the rows (ghost columns)
__global__ void dom_max_column(int n, const unsigned char *__restrict__ rowpat, const int *__restrict__ maxoff, int *__restrict__ out)
{
    __shared__ int part[4];
    int c = 0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) c = max(c, (int)r + maxoff[rowpat[r]]);
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(part[0], part[1]), max(part[2], part[3])));      // (one atomic per workgroup: two million of them on one address cost 40 ms at 512^3)
}
// plan time: what do the rows of each plane of the grid do with the neighbours their place in the grid puts OUTSIDE it?  desc[pattern]: 2 bits per slot of the dominant
// pattern (0 missing, 1 there with the dominant value, 2 there with the slot's alternative value), the count of trailing (row, +0.0) terms << 14, bit 20: none of that
// (foreign, values of its own).  A row is out (bad[z]) when a neighbour INSIDE the grid is not a plain dominant slot; for neighbours outside, obs[z] collects per kind
// (3 bits each) the states seen; bit 21: a row with missing slots and no padding terms, bit 22: one with a padding term per missing slot (other counts: bad).
__global__ void dom_box_check(int n, int S, int SO, int perm, const unsigned char *__restrict__ rowpat, const int *__restrict__ desc, int *__restrict__ bad, unsigned *__restrict__ obs)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int z = (int)(r / SO), rem = (int)(r - (long long)z * SO), yy = rem / S, xx = rem - yy * S, lines = SO / S, planes = n / SO;
    const int d = desc[rowpat[r]];
    bool b = ((d >> 20) & 1) != 0;
    unsigned o = 0;
    int missing = 0;
    for (int u = 0; u < 7; u++) {
        const int k = (perm >> (3 * u)) & 7, st = (d >> (2 * u)) & 3;
        const bool inside = k == 0 ? z > 0 : k == 1 ? yy > 0 : k == 2 ? xx > 0 : k == 3 ? true : k == 4 ? xx < S - 1 : k == 5 ? yy < lines - 1 : z < planes - 1;
        if (inside) { if (st != 1) b = true; }
        else { o |= 1u << (3 * k + st); if (st == 0) missing++; }
    }
    const int pads = (d >> 14) & 7;
    if (pads != 0 && pads != missing) b = true;
    if (missing > 0) o |= pads == 0 ? (1u << 21) : (1u << 22);
    if (b) bad[z] = 1;
    if (o & ~obs[z]) atomicOr(&obs[z], o);                     // (after the first rows of a plane everyone finds its bits there already)
}

// The dominant pattern of a plan with value records and the other patterns' records in ITS slots (spmv_csr_valuerec_dom_kernel).
// rec32: 8 ints per pattern (7 byte offsets, the tail repeating the last one; length), val8: 8 doubles per pattern.  Kept when
// one pattern carries at least half of the rows; never an error when it does not.
static void build_dominant(liship_csr_plan_s *p, int npat, const int *rec32, const double *val8 /* NULL: offsets only */)
{
    if (p->drec) { (void)hipFree(p->drec); p->drec = nullptr; }
    if (p->n < 4 * WAVE || npat <= 0 || npat > PAT7_MAX) return;
    unsigned long long *d_count = nullptr, count[256];
    if (hipMalloc(&d_count, sizeof(count)) != hipSuccess) return;
    bool ok = hipMemset(d_count, 0, sizeof(count)) == hipSuccess;
    if (ok) { rowpat_histogram<<<1024, 256>>>(p->n, p->rowpat, d_count); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpy(count, d_count, sizeof(count), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_count);
    if (!ok) return;
    int dom = 0;
    for (int i = 1; i < npat; i++) if (count[i] > count[dom]) dom = i;
    if (2 * count[dom] < (unsigned long long)p->n) return;
    const int *od = rec32 + 8 * dom;
    const int lend = od[7];
    DomRec D;
    for (int u = 0; u < 7; u++) { D.off[u] = od[u]; D.val[u] = val8 ? val8[8 * dom + u] : 0.0; }
    D.pat = dom; D.mask = (1 << lend) - 1; D.d0 = -1;
    for (int u = 0; u < lend; u++) if (od[u] == 0) D.d0 = u;
    const int d0 = D.d0;
    bool simple = true;
    double img[PAT7_MAX * 8];
    int desc[256];
    double alt[7] = {0, 0, 0, 0, 0, 0, 0};
    bool alt_set[7] = {false, false, false, false, false, false, false};
    for (int i = 0; i < 256; i++) desc[i] = 1 << 20;
    for (int i = 0; i < npat; i++) {
        const int *oi = rec32 + 8 * i;
        const int leni = oi[7];
        unsigned mask = 0;
        double *out = img + 8 * i;
        for (int u = 0; u < 8; u++) out[u] = 0.0;
        int j = 0;
        for (int sl = 0; sl < lend && j < leni; sl++)
            if (oi[j] == od[sl]) { mask |= 1u << sl; out[1 + sl] = val8 ? val8[8 * i + j] : 0.0; j++; }
        if (j != leni && val8 && d0 >= 0 && j >= 1) {     // ... followed by nothing but (row, +0.0) entries: ELL's padding (dom_pad_terms)
            bool pads = true;
            for (int k = j; k < leni; k++) { unsigned long long vb; memcpy(&vb, &val8[8 * i + k], 8); if (oi[k] != 0 || vb != 0ull) pads = false; }
            if (pads) { mask |= (unsigned)(leni - j) << 8; j = leni; }
        }
        if (j != leni) mask = 0x80u;                      // not a subsequence of the dominant pattern: its rows take their own records
        bool same = val8 != nullptr && mask < 0x80u;      // ... a plain mask whose kept slots carry the dominant pattern's values, bit for bit
        for (int sl = 0; sl < lend && same; sl++) if ((mask >> sl) & 1u) same = memcmp(&out[1 + sl], &D.val[sl], 8) == 0;
        simple = simple && same;
        if (same) mask |= 0x800u;                         // (bit 11: the dominant pattern's values in the slots it keeps)
        {   // the box check's view of the pattern
            int dsc = 0;
            bool complex_ = (mask & 0x80u) != 0 || val8 == nullptr;
            for (int sl = 0; sl < 7 && !complex_; sl++) {
                if (!((mask >> sl) & 1u)) continue;
                if (sl < lend && memcmp(&out[1 + sl], &D.val[sl], 8) == 0) { dsc |= 1 << (2 * sl); continue; }
                if (!alt_set[sl]) { alt_set[sl] = true; alt[sl] = out[1 + sl]; }
                if (memcmp(&out[1 + sl], &alt[sl], 8) != 0) complex_ = true;      // a third value in this slot
                dsc |= 2 << (2 * sl);
            }
            desc[i] = complex_ ? (1 << 20) : (dsc | (int)(((mask >> 8) & 7u) << 14));
        }
        unsigned long long bits = mask;
        memcpy(out, &bits, 8);
    }
    p->dom_simple = simple ? 1 : 0;
    int minoff = 0, maxoff = 0;                           // in elements
    for (int u = 0; u < lend; u++) { const int e = od[u] / 8; if (e < minoff) minoff = e; if (e > maxoff) maxoff = e; }
    if (hipMalloc(&p->drec, sizeof(double) * 8 * (size_t)npat) != hipSuccess) { p->drec = nullptr; return; }
    if (hipMemcpy(p->drec, img, sizeof(double) * 8 * (size_t)npat, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->drec); p->drec = nullptr; return; }
    p->dom = D;
    p->dom_lo = -minoff;
    p->dom_hi = p->n - maxoff;
    p->box_z0 = p->box_z1 = 0;
    p->dom_xlen = 0;
    if (p->rowpat) {
        int mo[256], *d_mo = nullptr, *d_out = nullptr, top = 0;
        for (int i = 0; i < 256; i++) mo[i] = 0;
        for (int i = 0; i < npat; i++) { const int *oi = rec32 + 8 * i; for (int u = 0; u < oi[7] && u < 7; u++) if (oi[u] / 8 > mo[i]) mo[i] = oi[u] / 8; }
        if (hipMalloc(&d_mo, sizeof(mo)) == hipSuccess && hipMalloc(&d_out, sizeof(int)) == hipSuccess && hipMemcpy(d_mo, mo, sizeof(mo), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemset(d_out, 0, sizeof(int)) == hipSuccess) {
            dom_max_column<<<2048, 256>>>(p->n, p->rowpat, d_mo, d_out);
            if (hipGetLastError() == hipSuccess && hipMemcpy(&top, d_out, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) p->dom_xlen = top + 1;
        }
        if (d_mo) (void)hipFree(d_mo);
        if (d_out) (void)hipFree(d_out);
    }
    int S = 0, SO = 0, perm = 0;
    bool finite = true;
    for (int u = 0; u < 7; u++) finite = finite && D.val[u] - D.val[u] == 0.0;      // (the BOX form's signed zeros need finite values)
    if (finite && val8 && p->rowpat && dom_seven_point(D, S, SO, perm) && p->n % SO == 0 && SO % S == 0) {
        const int planes = p->n / SO;
        int *d_bad = nullptr, *d_desc = nullptr, *bad = (int *)malloc(sizeof(int) * (size_t)planes);
        unsigned *d_obs = nullptr, *obs = (unsigned *)malloc(sizeof(unsigned) * (size_t)planes);
        if (bad && obs && hipMalloc(&d_bad, sizeof(int) * (size_t)planes) == hipSuccess && hipMalloc(&d_obs, sizeof(unsigned) * (size_t)planes) == hipSuccess &&
            hipMalloc(&d_desc, sizeof(desc)) == hipSuccess && hipMemcpy(d_desc, desc, sizeof(desc), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemset(d_bad, 0, sizeof(int) * (size_t)planes) == hipSuccess && hipMemset(d_obs, 0, sizeof(unsigned) * (size_t)planes) == hipSuccess) {
            dom_box_check<<<(p->n + 255) / 256, 256>>>(p->n, S, SO, perm, p->rowpat, d_desc, d_bad, d_obs);
            if (hipGetLastError() == hipSuccess && hipMemcpy(bad, d_bad, sizeof(int) * (size_t)planes, hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(obs, d_obs, sizeof(unsigned) * (size_t)planes, hipMemcpyDeviceToHost) == hipSuccess) {
                // the longest run of planes that agree: per kind of neighbour ONE way to treat it outside the grid, padding terms everywhere or nowhere, and not both
                // alternative values and padding terms (no kernel for that)
                auto agree = [](unsigned u) {
                    bool alts = false;
                    for (int k = 0; k < 7; k++) { const unsigned m3 = (u >> (3 * k)) & 7u; if (m3 & (m3 - 1)) return false; alts = alts || (m3 & 4u); }
                    const bool pn = (u >> 21) & 1u, py = (u >> 22) & 1u;
                    return !(pn && py) && !(py && alts);
                };
                int best0 = 0, best1 = 0; unsigned bestu = 0;
                for (int z = 0; z < planes; z++) {
                    if (bad[z] || !agree(obs[z]) || (best1 > z && z >= best0)) continue;      // (inside the best run so far: a run from here is shorter)
                    unsigned u = 0;
                    int e = z;
                    while (e < planes && !bad[e] && agree(u | obs[e])) { u |= obs[e]; e++; }
                    if (e - z > best1 - best0) { best0 = z; best1 = e; bestu = u; }
                }
                p->box_z0 = best0; p->box_z1 = best1;
                p->box_modes = 0;
                for (int k = 0; k < 7; k++) { const unsigned m3 = (bestu >> (3 * k)) & 7u; p->box_modes |= (m3 & 2u ? 1 : m3 & 4u ? 2 : 0) << (2 * k); }
                p->box_pads = (bestu >> 22) & 1u;
                for (int u = 0; u < 7; u++) p->box_alt[u] = alt[u];
            }
        }
        if (d_bad) (void)hipFree(d_bad);
        if (d_obs) (void)hipFree(d_obs);
        if (d_desc) (void)hipFree(d_desc);
        free(bad); free(obs);
    }
}

// the 96 B records (32 B offsets + length, 64 B values) of npat patterns -> p->vrec
static int install_value_records(liship_csr_plan_s *p, int npat, const int *rec32, const double *val8)
{
    unsigned char img[PAT7_MAX * 96];
    for (int i = 0; i < npat; i++) { memcpy(img + 96 * i, rec32 + 8 * i, 32); memcpy(img + 96 * i + 32, val8 + 8 * i, 64); }
    hipError_t e = hipMalloc(&p->vrec, 96 * (size_t)npat);
    if (e == hipSuccess) e = hipMemcpy(p->vrec, img, 96 * (size_t)npat, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (p->vrec) (void)hipFree(p->vrec); p->vrec = nullptr; return (int)e; }
    return 0;
}

// Split the offset patterns by the values their rows carry (see csr_collect_value_patterns).  On success the plan's pattern
// bytes, pattern table, 32 B records and value records are all replaced by the refined set (more patterns, some with the same
// offsets), so that every kernel that reads the pattern bytes keeps working; otherwise nothing changes.  old32: the 32 B records
// of the current patterns (host).
static int refine_patterns_by_values(liship_csr_plan_s *p, const int *ptr, const double *val, const int *old32, hipStream_t st)
{
    struct Table { unsigned long long keys[PAT_SLOTS]; int rep[PAT_SLOTS]; int count; };
    Table *host = (Table *)malloc(sizeof(Table)), *dev = nullptr;
    if (!host) return 0;
    memset(host, 0, sizeof(Table));
    for (int i = 0; i < PAT_SLOTS; i++) host->rep[i] = 0x7fffffff;
    int rc = 0;
#define PT(expr) do { if (rc == 0) { hipError_t e__ = (expr); if (e__ != hipSuccess) rc = (int)e__; } } while (0)
    PT(hipMalloc(&dev, sizeof(Table)));
    PT(hipMemcpyAsync(dev, host, sizeof(Table), hipMemcpyHostToDevice, st));
    if (rc == 0) { csr_collect_value_patterns<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, dev->keys, dev->rep, &dev->count); PT(hipGetLastError()); }
    PT(hipMemcpyAsync(host, dev, sizeof(Table), hipMemcpyDeviceToHost, st));
    PT(hipStreamSynchronize(st));
    if (dev) (void)hipFree(dev);
    if (rc != 0 || host->count <= 0 || host->count > PAT7_MAX) { free(host); return rc; }
    int npat = 0, order[PAT7_MAX];
    for (int i = 0; i < PAT_SLOTS && npat < PAT7_MAX; i++) if (host->keys[i] != 0ull) order[npat++] = i;
    for (int i = 1; i < npat; i++) { const int v = order[i]; int j = i - 1; while (j >= 0 && host->keys[order[j]] > host->keys[v]) { order[j + 1] = order[j]; j--; } order[j + 1] = v; }
    unsigned long long hashes[PAT7_MAX]; int reps[PAT7_MAX], oldpat[PAT7_MAX], bad = 1;
    double hv[PAT7_MAX * 8];
    for (int i = 0; i < npat; i++) { hashes[i] = host->keys[order[i]]; reps[i] = host->rep[order[i]]; }
    free(host);
    unsigned long long *d_hash = nullptr; int *d_rep = nullptr, *d_old = nullptr, *d_bad = nullptr; double *d_v = nullptr; unsigned char *newpat = nullptr;
    PT(hipMalloc(&d_hash, sizeof(hashes))); PT(hipMalloc(&d_rep, sizeof(reps))); PT(hipMalloc(&d_old, sizeof(oldpat)));
    PT(hipMalloc(&d_bad, sizeof(int))); PT(hipMalloc(&d_v, sizeof(hv))); PT(hipMalloc(&newpat, (size_t)p->n + 64));
    PT(hipMemcpyAsync(d_hash, hashes, sizeof(unsigned long long) * npat, hipMemcpyHostToDevice, st));
    PT(hipMemcpyAsync(d_rep, reps, sizeof(int) * npat, hipMemcpyHostToDevice, st));
    PT(hipMemsetAsync(d_bad, 0, sizeof(int), st));
    if (rc == 0) { csr_fetch_value_patterns<<<1, 64, 0, st>>>(npat, d_rep, ptr, val, p->rowpat, d_old, d_v); PT(hipGetLastError()); }
    if (rc == 0) {
        csr_encode_value_patterns<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, npat, d_hash, d_old, d_v, newpat, d_bad);
        PT(hipGetLastError());
    }
    PT(hipMemcpyAsync(oldpat, d_old, sizeof(int) * npat, hipMemcpyDeviceToHost, st));
    PT(hipMemcpyAsync(hv, d_v, sizeof(double) * 8 * npat, hipMemcpyDeviceToHost, st));
    PT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    PT(hipStreamSynchronize(st));
    (void)hipFree(d_hash); (void)hipFree(d_rep); (void)hipFree(d_old); (void)hipFree(d_bad); (void)hipFree(d_v);
    // the refined tables: pattern i has the offsets of offset pattern oldpat[i]
    int rec[PAT7_MAX * 8], tab[PAT7_MAX + 1 + PAT7_MAX * 7], total = 0, *d_tab = nullptr; v4i32 *d_rec = nullptr;
    if (rc == 0 && bad == 0) {
        for (int i = 0; i < npat; i++) {
            memcpy(rec + 8 * i, old32 + 8 * oldpat[i], 32);
            tab[i] = total;
            total += rec[8 * i + 7];
        }
        tab[npat] = total;
        int at = npat + 1;
        for (int i = 0; i < npat; i++) for (int j = 0; j < rec[8 * i + 7]; j++) tab[at++] = rec[8 * i + j] / 8;     // the general table holds element offsets
        PT(hipMalloc(&d_tab, sizeof(int) * (size_t)(npat + 1 + total)));
        PT(hipMalloc(&d_rec, sizeof(int) * 8 * (size_t)npat));
        PT(hipMemcpy(d_tab, tab, sizeof(int) * (size_t)(npat + 1 + total), hipMemcpyHostToDevice));
        PT(hipMemcpy(d_rec, rec, sizeof(int) * 8 * (size_t)npat, hipMemcpyHostToDevice));
    }
#undef PT
    if (rc != 0 || bad != 0) { if (newpat) (void)hipFree(newpat); if (d_tab) (void)hipFree(d_tab); if (d_rec) (void)hipFree(d_rec); return rc; }
    rc = install_value_records(p, npat, rec, hv);
    if (rc != 0) { (void)hipFree(newpat); (void)hipFree(d_tab); (void)hipFree(d_rec); return rc; }
    (void)hipFree(p->rowpat); (void)hipFree(p->ptab); (void)hipFree(p->ptab8);
    p->rowpat = newpat; p->ptab = d_tab; p->ptab8 = d_rec;
    p->npat = npat; p->ptab_len = npat + 1 + total;
    for (int i = 0; i < npat; i++) p->prep[i] = reps[i];
    build_dominant(p, npat, rec, hv);               // (on the refined pattern bytes)
    return 0;
}

// Value records on top of the pattern records (see spmv_csr_valuerec_kernel): setup-time, optional, never an error when the matrix
// does not qualify (no 32 B records, or two rows of one pattern with different values).  One pass over ptr / value.
// the same for patterns of up to 32 entries (no 32 B records): offsets from the plan's pattern table, values from one row per pattern,
// every row checked; the image is npat x 144 B (32 byte offsets, the tail repeating the last one; the length; padding) followed by
// npat x 256 B (32 values, the tail 0).  Rows of one offset pattern with different values split the pattern (up to 48 in all).
// The dominant pattern of a plan with WIDE value records, its runs of neighbouring columns as 64-row slots, every other pattern as a mask and values in the
// dominant one's slots (spmv_csr_valuerecw_staged_kernel).  T: the pattern table (NP + 1 prefix entries, then element offsets), vals: NP x PATW_LEN
// values (host; NULL for a plan whose values are streamed: masks and slots only, spmv_csr_pattern_rows_staged_kernel).  Kept when one pattern carries at
// least half of the rows; never an error.
// a shortest common supersequence of two patterns (offsets, values), an entry of one being an entry of the other when offset and value bits agree; <= 2 * PATW_LEN long
static int scs_merge(const int *ao, const double *av, int al, const int *bo, const double *bv, int bl, int *mo, double *mv)
{
    static_assert(PATW_LEN < 255, "lengths in a byte");
    unsigned char L[PATW_LEN + 1][PATW_LEN + 1];                // L[a][b]: the longest common subsequence of the tails
    auto eq = [&](int a, int b) { return ao[a] == bo[b] && memcmp(&av[a], &bv[b], 8) == 0; };
    for (int a = al; a >= 0; a--)
        for (int b = bl; b >= 0; b--)
            L[a][b] = (a == al || b == bl) ? 0 : eq(a, b) ? (unsigned char)(1 + L[a + 1][b + 1]) : (L[a + 1][b] >= L[a][b + 1] ? L[a + 1][b] : L[a][b + 1]);
    int a = 0, b = 0, ml = 0;
    while (a < al || b < bl) {
        if (a < al && b < bl && eq(a, b)) { mo[ml] = ao[a]; mv[ml++] = av[a]; a++; b++; }
        else if (b >= bl || (a < al && L[a + 1][b] >= L[a][b + 1])) { mo[ml] = ao[a]; mv[ml++] = av[a]; a++; }
        else { mo[ml] = bo[b]; mv[ml++] = bv[b]; b++; }
    }
    return ml;
}

// Is the plan the 27-point box stencil the marching kernel serves (spmv_csr_box27_march_kernel)?  The dominant pattern's offsets are dz SO + dy S + dx in ascending
// order, lines a multiple of 128 long, a multiple of four lines per plane, whole planes; the values finite (the zero that stands in for a neighbour outside the grid
// turns each of them into a +-0.0 term, and a sum that starts at +0.0 is never -0.0, so such terms cannot change its bits); and every row keeps exactly the slots
// whose neighbour lies inside the grid, with the dominant values (wide_box_check, one pass over the pattern bytes).
static void try_box27(liship_csr_plan_s *p, const int *od, const WideDom &D, hipStream_t st)
{
    p->b27.S = 0;
    if (D.len != 27 || D.pat < 0 || !p->wdrec || !p->rowpat) return;
    const int S = od[16], SO = od[22];
    if (S < 128 || S % 2 != 0 || (S % 128 != 0 && S % 128 < 4) || SO < 4 * S || SO % S != 0 || p->n % SO != 0 || p->n / SO < 2) return;      // (lines of any even length from 128 on, any number of lines from 4 on: partial tiles)
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++)
        if (od[(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] != dz * SO + dy * S + dx) return;
    for (int u = 0; u < 27; u++) {
        const double v = D.val[u];
        if (!(v == v) || v - v != 0.0) return;                        // NaN, infinite: their product with the halo's zero would not be a zero
    }
    int *d_bad = nullptr, bad = 1;
    if (hipMalloc(&d_bad, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return; }
    bool ok = hipMemsetAsync(d_bad, 0, sizeof(int), st) == hipSuccess;
    if (ok) { wide_box_check<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, S, SO, p->rowpat, p->wdrec, d_bad); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(d_bad);
    if (!ok || bad != 0) return;
    Box27 B;
    memset(&B, 0, sizeof(B));
    B.S = S; B.SO = SO; B.planes = p->n / SO;
    B.poison = 0.0;                                                   // (finite value) * 0.0 = +-0.0: a term that cannot change a sum that started at +0.0 (box27_shape asks for that)
    for (int u = 0; u < 27; u++) B.val[u] = D.val[u];
    p->b27 = B;
}

static void build_wide_dominant(liship_csr_plan_s *p, const int *T, int NP, const double *vals, const int *ptr, hipStream_t st)
{
    p->b27.S = 0;
    if (p->wdrec) { (void)hipFree(p->wdrec); p->wdrec = nullptr; }
    if (p->wstage) { (void)hipFree(p->wstage); p->wstage = nullptr; }
    p->wd.len = 0;
    if (NP <= 0 || NP > 255 || p->n < 4 * WAVE || !p->rowpat || !p->codes || !p->dict) return;
    unsigned long long *d_count = nullptr, count[256];
    int *d_max = nullptr, maxcol = 0;
    bool ok = hipMalloc(&d_count, sizeof(count)) == hipSuccess && hipMalloc(&d_max, sizeof(int)) == hipSuccess;
    ok = ok && hipMemsetAsync(d_count, 0, sizeof(count), st) == hipSuccess && hipMemsetAsync(d_max, 0, sizeof(int), st) == hipSuccess;
    if (ok) { rowpat_histogram<<<1024, 256, 0, st>>>(p->n, p->rowpat, d_count); ok = hipGetLastError() == hipSuccess; }
    if (ok) { csr_max_column<<<2048, 256, 0, st>>>(p->n, ptr, p->codes, p->dict, d_max); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(count, d_count, sizeof(count), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(&maxcol, d_max, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (d_count) (void)hipFree(d_count);
    if (d_max) (void)hipFree(d_max);
    if (!ok) return;
    int dom = 0;
    for (int i = 1; i < NP; i++) if (count[i] > count[dom]) dom = i;
    int l = T[dom + 1] - T[dom];
    if (l < 1 || l > PATW_LEN || maxcol < 1) return;
    const int *od = T + NP + 1 + T[dom];                       // the dominant pattern's offsets (elements), in its own order = slot order
    int uoff[PATW_LEN];
    double uval[PATW_LEN];
    if (count[dom] * 2 < (unsigned long long)p->n) {
        // No single pattern carries half of the rows (a b x b blocking of a stencil kept row by row has b interior patterns that take turns; lis_matrix_convert_csr2bsr
        // keeps a block row's blocks in first-seen order, so the turns do not even agree on the order of the columns they share): a VIRTUAL dominant pattern, a common
        // supersequence of the most frequent patterns -- entries are the same when offset and value bits are; an offset may sit in it twice, both entries reading one
        // staged slot -- so that each of them is an order-preserving mask over it with ITS values.  No row is ON it (D.pat = -1): every row takes the kernel's masked
        // scalar-register path, -0.0 for the entries it does not have, its own terms in its own order.  Value records only (the streamed-value kernels count kept slots).
        if (!vals || g_wide_union == 0 || (p->n < (1 << 19) && g_wide_union != 2)) return;       // (below half a million rows the gathering kernel's shorter chain of round trips wins: 64^3 0.0073 against 0.0094 ms; 2 = always, for tests)
        int order[256], ul = 0;
        unsigned long long covered = 0;
        for (int i = 0; i < NP; i++) order[i] = i;
        for (int a = 1; a < NP; a++) { const int v = order[a]; int b = a - 1; while (b >= 0 && count[order[b]] < count[v]) { order[b + 1] = order[b]; b--; } order[b + 1] = v; }
        for (int t = 0; t < NP && count[order[t]] * 64 >= (unsigned long long)p->n; t++) {
            const int i = order[t], li = T[i + 1] - T[i];
            if (li < 1 || li > PATW_LEN) continue;
            int moff[2 * PATW_LEN];
            double mval[2 * PATW_LEN];
            const int ml = scs_merge(uoff, uval, ul, T + NP + 1 + T[i], vals + (size_t)i * PATW_LEN, li, moff, mval);
            // a longer supersequence is more masked work for EVERY row: grown only for a pattern with an eighth of the rows, or while half of them are not covered yet
            if (ml > PATW_LEN || (ml > ul && ul > 0 && count[i] * 8 < (unsigned long long)p->n && covered * 2 >= (unsigned long long)p->n)) continue;
            memcpy(uoff, moff, sizeof(int) * (size_t)ml); memcpy(uval, mval, sizeof(double) * (size_t)ml);
            ul = ml; covered += count[i];
        }
        if (ul < 1 || covered * 2 < (unsigned long long)p->n) return;
        dom = -1; l = ul; od = uoff;
    }
    int offs[PATW_LEN], start[16], mlen[16], base[17], nruns = 0;
    for (int j = 0; j < l; j++) offs[j] = od[j];
    for (int a = 1; a < l; a++) { const int v = offs[a]; int b = a - 1; while (b >= 0 && offs[b] > v) { offs[b + 1] = offs[b]; b--; } offs[b + 1] = v; }
    int lu = l;                                                 // distinct offsets (the virtual pattern may hold one twice: one staged slot)
    if (dom < 0) { lu = 0; for (int j = 0; j < l; j++) if (lu == 0 || offs[j] != offs[lu - 1]) offs[lu++] = offs[j]; }
    for (int j = 0; j < lu; ) {
        int e = j + 1;
        while (e < lu && offs[e] == offs[e - 1] + 1) e++;
        if ((e < lu && offs[e] == offs[e - 1]) || nruns == 16) return;     // a repeated offset, too many runs: the gathering kernel serves
        start[nruns] = offs[j]; mlen[nruns++] = e - j;
        j = e;
    }
    base[0] = 0;
    for (int a = 0; a < nruns; a++) base[a + 1] = base[a] + ((WAVE - 1 + mlen[a] + 1) & ~1);     // 64 rows need 63 + m columns of a run of m; even widths (pairs of slots)
    const int slots = base[nruns];
    if (slots > 8 * 2 * WAVE) return;
    WideDom D;
    memset(&D, 0, sizeof(D));
    D.len = l; D.pat = dom; D.slots = slots; D.maxcol = maxcol;
    for (int j = 0; j < l; j++) {
        int q = 0;
        for (int a = 0; a < nruns; a++) if (od[j] >= start[a] && od[j] < start[a] + mlen[a]) q = a;
        D.slot[j] = base[q] + (od[j] - start[q]);
        D.val[j] = dom < 0 ? uval[j] : vals ? vals[(size_t)dom * PATW_LEN + j] : 0.0;
    }
    D.tri = (l % 3 == 0 && l <= 30) ? 1 : 0;
    for (int q = 0; q < l / 3 && D.tri; q++) if (D.slot[3 * q + 1] != D.slot[3 * q] + 1 || D.slot[3 * q + 2] != D.slot[3 * q] + 2) D.tri = 0;
    int *stage = (int *)calloc(WAVE * 8, sizeof(int));
    double *img = (double *)calloc((size_t)NP * WREC, sizeof(double));
    if (stage && img) {
        for (int lane = 0; lane < WAVE; lane++)
            for (int k = 0; k < 8; k++) {
                const int sl = 2 * (k * WAVE + lane);
                int q = -1;
                for (int a = 0; a < nruns; a++) if (sl >= base[a] && sl < base[a + 1]) q = a;
                stage[lane * 8 + k] = q >= 0 ? start[q] + (sl - base[q]) : 0;
            }
        for (int i = 0; i < NP; i++) {
            const int li = T[i + 1] - T[i];
            const int *oi = T + NP + 1 + T[i];
            unsigned long long bits = 0;
            int j = 0;
            if (dom < 0) {                                    // the virtual pattern: first as a mask over entries that carry this pattern's VALUES too (leftmost match)
                for (int sl = 0; sl < l && j < li; sl++)
                    if (oi[j] == od[sl] && memcmp(&vals[(size_t)i * PATW_LEN + j], &D.val[sl], 8) == 0) { bits |= 1ull << sl; j++; }
                if (j != li) { bits = 0; j = 0; }
                else for (int sl = 0; sl < l; sl++) if ((bits >> sl) & 1ull) img[(size_t)i * WREC + sl] = D.val[sl];
            }
            for (int sl = 0; sl < l && j < li; sl++)
                if (oi[j] == od[sl]) { bits |= 1ull << sl; img[(size_t)i * WREC + sl] = vals ? vals[(size_t)i * PATW_LEN + j] : 0.0; j++; }
            if (j != li) bits = 1ull << 32;                   // not a subsequence of the dominant pattern: its rows walk their own record
            else {                                            // every kept slot carries the dominant pattern's value: the kernel leaves the values in scalar registers
                bool same = true;
                for (int sl = 0; sl < l; sl++) if ((bits >> sl) & 1ull) same = same && memcmp(&img[(size_t)i * WREC + sl], &D.val[sl], 8) == 0;
                if (same && vals) bits |= 1ull << 33;
            }
            memcpy(&img[(size_t)i * WREC + 32], &bits, 8);
        }
        if (hipMalloc(&p->wdrec, sizeof(double) * WREC * (size_t)NP) == hipSuccess && hipMalloc(&p->wstage, sizeof(int) * WAVE * 8) == hipSuccess &&
            hipMemcpy(p->wdrec, img, sizeof(double) * WREC * (size_t)NP, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(p->wstage, stage, sizeof(int) * WAVE * 8, hipMemcpyHostToDevice) == hipSuccess) { p->wd = D; if (vals && dom >= 0) try_box27(p, od, D, st); }
        else {
            if (p->wdrec) { (void)hipFree(p->wdrec); p->wdrec = nullptr; }
            if (p->wstage) { (void)hipFree(p->wstage); p->wstage = nullptr; }
        }
    }
    free(stage); free(img);
}

static int encode_wide_value_records(liship_csr_plan_s *p, const int *ptr, const double *val, hipStream_t st)
{
    if (p->npat <= 0 || p->npat > PATW_MAX || p->ptab_len <= p->npat) return 0;
    const int npat = p->npat;
    int *tab = (int *)malloc(sizeof(int) * (size_t)p->ptab_len);
    if (!tab) return 0;
    int rc = 0, bad = 1;
#define PT(expr) do { if (rc == 0) { hipError_t e__ = (expr); if (e__ != hipSuccess) rc = (int)e__; } } while (0)
    PT(hipMemcpy(tab, p->ptab, sizeof(int) * (size_t)p->ptab_len, hipMemcpyDeviceToHost));
    int maxlen = 0, minlen = 1 << 30, maxoff = 0;
    for (int i = 0; i < npat && rc == 0; i++) { const int l = tab[i + 1] - tab[i]; if (l > maxlen) maxlen = l; if (l < minlen) minlen = l; }
    for (int t = npat + 1; t < p->ptab_len && rc == 0; t++) if (tab[t] > maxoff) maxoff = tab[t];
    if (rc != 0 || maxlen > PATW_LEN || minlen < 1 || (long long)p->n + maxoff >= (1ll << 29)) { free(tab); return rc; }
    int *d_rep = nullptr, *d_bad = nullptr; double *vr = nullptr;
    PT(hipMalloc(&d_rep, sizeof(int) * (size_t)npat)); PT(hipMalloc(&d_bad, sizeof(int))); PT(hipMalloc(&vr, sizeof(double) * PATW_LEN * (size_t)npat));
    PT(hipMemcpyAsync(d_rep, p->prep, sizeof(int) * (size_t)npat, hipMemcpyHostToDevice, st));
    PT(hipMemsetAsync(d_bad, 0, sizeof(int), st));
    if (rc == 0) { csr_fetch_values<<<(npat + 63) / 64, 64, 0, st>>>(npat, d_rep, ptr, val, vr, PATW_LEN, PATW_LEN); PT(hipGetLastError()); }
    if (rc == 0) { csr_check_values<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, vr, d_bad, PATW_LEN, PATW_LEN); PT(hipGetLastError()); }
    PT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    PT(hipStreamSynchronize(st));
    // rows of one offset pattern with different values: split the patterns by (pattern, values), as for the 7-entry records
    int np2 = npat, *oldpat = nullptr, *ntab = nullptr, ntab_len = 0, reps2[PATW_MAX];
    unsigned char *newpat = nullptr;
    double *vr2 = nullptr;                               // device: the refined patterns' values
    if (rc == 0 && bad != 0) {
        struct Table { unsigned long long keys[PAT_SLOTS]; int rep[PAT_SLOTS]; int count; };
        Table *host = (Table *)calloc(1, sizeof(Table)), *dev = nullptr;
        if (host) {
            for (int i = 0; i < PAT_SLOTS; i++) host->rep[i] = 0x7fffffff;
            PT(hipMalloc(&dev, sizeof(Table)));
            PT(hipMemcpyAsync(dev, host, sizeof(Table), hipMemcpyHostToDevice, st));
            if (rc == 0) { csr_collect_value_patterns<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, dev->keys, dev->rep, &dev->count); PT(hipGetLastError()); }
            PT(hipMemcpyAsync(host, dev, sizeof(Table), hipMemcpyDeviceToHost, st));
            PT(hipStreamSynchronize(st));
            if (dev) (void)hipFree(dev);
            if (rc == 0 && host->count > 0 && host->count <= PATW_MAX) {
                int order[PATW_MAX];
                np2 = 0;
                for (int i = 0; i < PAT_SLOTS && np2 < PATW_MAX; i++) if (host->keys[i] != 0ull) order[np2++] = i;
                for (int i = 1; i < np2; i++) { const int v = order[i]; int j = i - 1; while (j >= 0 && host->keys[order[j]] > host->keys[v]) { order[j + 1] = order[j]; j--; } order[j + 1] = v; }
                unsigned long long hashes[PATW_MAX];
                for (int i = 0; i < np2; i++) { hashes[i] = host->keys[order[i]]; reps2[i] = host->rep[order[i]]; }
                unsigned long long *d_hash = nullptr; int *d_rep2 = nullptr, *d_old = nullptr;
                oldpat = (int *)malloc(sizeof(int) * PATW_MAX);
                PT(hipMalloc(&d_hash, sizeof(hashes))); PT(hipMalloc(&d_rep2, sizeof(int) * PATW_MAX)); PT(hipMalloc(&d_old, sizeof(int) * PATW_MAX));
                PT(hipMalloc(&vr2, sizeof(double) * PATW_LEN * PATW_MAX)); PT(hipMalloc(&newpat, (size_t)p->n + 64));
                PT(hipMemcpyAsync(d_hash, hashes, sizeof(unsigned long long) * np2, hipMemcpyHostToDevice, st));
                PT(hipMemcpyAsync(d_rep2, reps2, sizeof(int) * np2, hipMemcpyHostToDevice, st));
                PT(hipMemsetAsync(d_bad, 0, sizeof(int), st));
                if (rc == 0) { csr_fetch_value_patterns<<<(np2 + 63) / 64, 64, 0, st>>>(np2, d_rep2, ptr, val, p->rowpat, d_old, vr2, PATW_LEN, PATW_LEN); PT(hipGetLastError()); }
                if (rc == 0) {
                    csr_encode_value_patterns<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, np2, d_hash, d_old, vr2, newpat, d_bad, PATW_LEN, PATW_LEN);
                    PT(hipGetLastError());
                }
                if (oldpat) PT(hipMemcpyAsync(oldpat, d_old, sizeof(int) * np2, hipMemcpyDeviceToHost, st));
                PT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
                PT(hipStreamSynchronize(st));
                (void)hipFree(d_hash); (void)hipFree(d_rep2); (void)hipFree(d_old);
                if (rc == 0 && bad == 0 && oldpat) {         // the general pattern table of the refined set (prefix, then element offsets)
                    int total = 0;
                    for (int i = 0; i < np2; i++) total += tab[oldpat[i] + 1] - tab[oldpat[i]];
                    ntab_len = np2 + 1 + total;
                    ntab = ntab_len <= PAT_TABLE ? (int *)malloc(sizeof(int) * (size_t)ntab_len) : nullptr;     // (the general pattern kernel keeps the table in LDS)
                    if (ntab) {
                        int at = 0;
                        for (int i = 0; i < np2; i++) {
                            const int l = tab[oldpat[i] + 1] - tab[oldpat[i]];
                            ntab[i] = at;
                            for (int j = 0; j < l; j++) ntab[np2 + 1 + at + j] = tab[npat + 1 + tab[oldpat[i]] + j];
                            at += l;
                        }
                        ntab[np2] = at;
                    } else bad = 1;
                } else bad = 1;
            }
            free(host);
        }
    }
    const int *T = ntab ? ntab : tab;                    // the table the records are built from, and its pattern count
    const int NP = ntab ? np2 : npat;
    const size_t obytes = sizeof(int) * PATW_OFF * (size_t)NP, vbytes = sizeof(double) * PATW_LEN * (size_t)NP;
    unsigned char *img = (unsigned char *)calloc(1, obytes + vbytes);
    if (rc == 0 && bad == 0 && img) {
        int *off = (int *)img;
        for (int i = 0; i < NP; i++) {
            const int l = T[i + 1] - T[i];
            for (int j = 0; j < PATW_LEN; j++) off[PATW_OFF * i + j] = 8 * T[NP + 1 + T[i] + (j < l ? j : l - 1)];
            off[PATW_OFF * i + PATW_LEN] = l;
        }
        PT(hipMemcpy(img + obytes, ntab ? vr2 : vr, vbytes, hipMemcpyDeviceToHost));
        PT(hipMalloc(&p->vrecw, obytes + vbytes));
        PT(hipMemcpy(p->vrecw, img, obytes + vbytes, hipMemcpyHostToDevice));
        int *d_tab = nullptr;
        if (ntab) {                                      // the refined pattern bytes and table replace the plan's
            PT(hipMalloc(&d_tab, sizeof(int) * (size_t)ntab_len));
            PT(hipMemcpy(d_tab, ntab, sizeof(int) * (size_t)ntab_len, hipMemcpyHostToDevice));
        }
        if (rc != 0) { if (p->vrecw) (void)hipFree(p->vrecw); p->vrecw = nullptr; if (d_tab) (void)hipFree(d_tab); }
        else if (ntab) {
            (void)hipFree(p->rowpat); (void)hipFree(p->ptab);
            p->rowpat = newpat; newpat = nullptr; p->ptab = d_tab; p->npat = np2; p->ptab_len = ntab_len;
            for (int i = 0; i < np2; i++) p->prep[i] = reps2[i];
            build_team_records(p, ntab, np2);            // (the values-streamed product of this plan reads the renumbered pattern bytes too)
            build_team_runs(p, ptr, st);
        }
        if (rc == 0 && p->vrecw) build_wide_dominant(p, T, NP, reinterpret_cast<const double *>(img + obytes), ptr, st);
    }
#undef PT
    (void)hipFree(d_rep); (void)hipFree(d_bad); (void)hipFree(vr);
    if (vr2) (void)hipFree(vr2);
    if (newpat) (void)hipFree(newpat);
    free(img); free(tab); free(ntab); free(oldpat);
    return rc;
}

extern "C" int liship_csr_plan_encode_row_values(liship_csr_plan_t p, const int *ptr, const double *val, void *stream)
{
    if (!p || (p->n > 0 && (!ptr || !val))) return LISHIP_ERR_ARG;
    if (p->rowpat && !p->ptab8 && !p->vrecw && !p->vrec && g_variant == 0) return encode_wide_value_records(p, ptr, val, as_stream(stream));
    if (!p->rowpat || !p->ptab8 || p->vrec || p->npat <= 0 || p->npat > PAT7_MAX || g_variant != 0) return 0;
    hipStream_t st = as_stream(stream);
    int *d_rep = nullptr, *d_bad = nullptr, bad = 1, rc = 0;
    double *vr = nullptr;
#define PT(expr) do { if (rc == 0) { hipError_t e__ = (expr); if (e__ != hipSuccess) rc = (int)e__; } } while (0)
    PT(hipMalloc(&d_rep, sizeof(int) * (size_t)p->npat)); PT(hipMalloc(&d_bad, sizeof(int)));
    PT(hipMalloc(&vr, sizeof(double) * 8 * (size_t)p->npat));
    PT(hipMemcpyAsync(d_rep, p->prep, sizeof(int) * (size_t)p->npat, hipMemcpyHostToDevice, st));
    PT(hipMemsetAsync(d_bad, 0, sizeof(int), st));
    if (rc == 0) { csr_fetch_values<<<1, 64, 0, st>>>(p->npat, d_rep, ptr, val, vr); PT(hipGetLastError()); }
    if (rc == 0) { csr_check_values<<<(p->n + 255) / 256, 256, 0, st>>>(p->n, ptr, val, p->rowpat, vr, d_bad); PT(hipGetLastError()); }
    PT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    PT(hipStreamSynchronize(st));
#undef PT
    (void)hipFree(d_rep); (void)hipFree(d_bad);
    double hv[PAT7_MAX * 8];
    int hr[PAT7_MAX * 8];
    if (rc == 0) {
        hipError_t e = hipMemcpy(hv, vr, sizeof(double) * 8 * (size_t)p->npat, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hr, p->ptab8, sizeof(int) * 8 * (size_t)p->npat, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = (int)e;
    }
    if (vr) (void)hipFree(vr);
    if (rc != 0) return rc;
    if (bad != 0) return refine_patterns_by_values(p, ptr, val, hr, st);     // rows of one offset pattern with different values
    rc = install_value_records(p, p->npat, hr, hv);
    if (rc == 0) build_dominant(p, p->npat, hr, hv);
    return rc;
}
// 1 (2: wide records) when the plan keeps the rows' values in the pattern records (the products then read neither values nor indices), else 0
extern "C" int liship_csr_plan_value_records(liship_csr_plan_t p)
{ return (p && p->rowpat && p->ptab8 && p->vrec) ? 1 : (p && p->rowpat && p->vrecw) ? 2 : 0; }        // 2: the wide records (rows of up to 32 entries)
extern "C" int liship_spmv_csr_set_row_values(int on) { g_row_values = on ? 1 : 0; return 0; }

// plan time: the pattern byte of a block row's first row must determine its other rows' (tab: 256 x 3, 0xffffffff = not seen); block rows per first-row pattern
__global__ void blockrow_keys(int nbr, int b, const unsigned char *__restrict__ rowpat, unsigned int *__restrict__ tab, unsigned long long *__restrict__ count, int *__restrict__ bad)
{
    __shared__ unsigned int h[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) h[t] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nbr; i += (long long)gridDim.x * blockDim.x) {
        const int key = rowpat[i * b];
        atomicAdd(&h[key], 1u);
        for (int k = 1; k < b; k++) {
            const unsigned int v = rowpat[i * b + k], old = atomicCAS(&tab[key * 3 + k - 1], 0xffffffffu, v);
            if (old != 0xffffffffu && old != v) *bad = 1;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 256; t += blockDim.x) if (h[t]) atomicAdd(&count[t], (unsigned long long)h[t]);
}

// Block rows on top of the wide value records (see spmv_csr_blockrows_staged_kernel): setup-time, optional, never an error when the matrix does not qualify.
// b: the rows b i .. b i + b - 1 list the same columns in the same order for every i (the row form of a b x b BSR matrix, liship_bsr_to_rows).
// Is the block-row plan the 7-point stencil in 2 x 2 blocks on a box grid (spmv_csr_block2_march_kernel)?  The dominant block row's entries are seven blocks of two
// columns at block offsets {-SO, -S, -2, 0, +2, +S, +SO} (rows), both columns of a block adjacent in the list, lines a multiple of 128 long, a multiple of eight lines
// per plane, finite values; and every block row keeps exactly the blocks that lie inside the grid (block2_box_check).
static void try_block2_march(liship_csr_plan_s *p, const int *doff, const BlockDom &D, hipStream_t st)
{
    p->b2.S = 0;
    if (D.b != 2 || D.len != 14 || !p->bdrec || !p->rowpat) return;
    int pos[7], npos = 0;
    for (int j = 0; j < 14; j++) { const int o = doff[j] & ~1; if (o > 0) { bool seen = false; for (int k = 0; k < npos; k++) seen = seen || pos[k] == o; if (!seen && npos < 7) pos[npos++] = o; } }
    if (npos != 3) return;
    for (int a = 1; a < 3; a++) { const int v = pos[a]; int c = a - 1; while (c >= 0 && pos[c] > v) { pos[c + 1] = pos[c]; c--; } pos[c + 1] = v; }
    const int S = pos[1], SO = pos[2];
    if (pos[0] != 2 || S < 128 || S % 2 != 0 || (S % 128 != 0 && S % 128 < 4) || SO < 8 * S || SO % S != 0 || p->n % SO != 0 || p->n / SO < 2) return;      // (partial tiles: lines of any even length from 128 on, any number of lines from 8 on)
    Block2March M;
    memset(&M, 0, sizeof(M));
    M.S = S; M.SO = SO; M.planes = p->n / SO;
    int kseq[7];
    for (int j = 0; j < 14; j++) {
        const int o = doff[j], blk = o & ~1, c = o & 1;          // (two's complement: o & ~1 is the block's first column for negative offsets too)
        const int k = blk == -SO ? 0 : blk == -S ? 1 : blk == -2 ? 2 : blk == 0 ? 3 : blk == 2 ? 4 : blk == S ? 5 : blk == SO ? 6 : -1;
        if (k < 0 || c != (j & 1)) return;                       // a block's two columns are neighbours in the list, first column first
        if ((j & 1) == 0) kseq[j >> 1] = k; else if (kseq[j >> 1] != k) return;
        M.v0[j] = D.val[0][j]; M.v1[j] = D.val[1][j];
        if (!(M.v0[j] == M.v0[j]) || M.v0[j] - M.v0[j] != 0.0 || !(M.v1[j] == M.v1[j]) || M.v1[j] - M.v1[j] != 0.0) return;      // NaN / infinite
    }
    const int ord0[7] = {0, 1, 2, 3, 4, 5, 6}, ord1[7] = {0, 1, 2, 3, 5, 6, 4}, ord2[7] = {0, 6, 1, 5, 2, 3, 4};
    if (memcmp(kseq, ord0, sizeof(kseq)) == 0) M.ord = 0; else if (memcmp(kseq, ord1, sizeof(kseq)) == 0) M.ord = 1;
    else if (memcmp(kseq, ord2, sizeof(kseq)) == 0) M.ord = 2; else return;      // (other orders: the staged kernel)
    int *d_bad = nullptr, bad = 1;
    if (hipMalloc(&d_bad, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return; }
    bool ok = hipMemsetAsync(d_bad, 0, sizeof(int), st) == hipSuccess;
    const int nbr = p->n / 2;
    if (ok) { block2_box_check<<<(nbr + 255) / 256, 256, 0, st>>>(nbr, S, SO, 14, p->rowpat, p->bdrec, M, d_bad); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(d_bad);
    if (ok && bad == 0) p->b2 = M;
}

extern "C" int liship_csr_plan_encode_block_rows(liship_csr_plan_t p, int b, const int *ptr, void *stream)
{
    if (!p || b < 2 || b > 4 || (p->n > 0 && !ptr)) return LISHIP_ERR_ARG;
    if (!p->rowpat || !p->vrecw || p->ptab8 || !p->ptab || p->npat <= 0 || p->npat > 255 || p->n % b != 0 || p->n < 4 * WAVE * b || p->bd.len > 0 ||
        !p->codes || !p->dict || g_variant != 0) return 0;
    if (p->n < (1 << 17) && g_block_rows != 2) return 0;            // (small matrices: the gathering kernel's shorter chain of round trips wins, 2 x 2 at 32^3: 0.0035 against 0.0052 ms; from 48^3 on it does not)
    hipStream_t st = as_stream(stream);
    const int NP = p->npat, nbr = p->n / b;
    const size_t obytes = sizeof(int) * PATW_OFF * (size_t)NP, vbytes = sizeof(double) * PATW_LEN * (size_t)NP;
    int *T = (int *)malloc(sizeof(int) * (size_t)p->ptab_len);
    unsigned char *img = (unsigned char *)malloc(obytes + vbytes);
    unsigned int *d_tab = nullptr, tab[256 * 3];
    unsigned long long *d_count = nullptr, count[256];
    int *d_bad = nullptr, bad[2] = {1, 0};                       // {rows of one first-row pattern disagree, the largest column}
    bool ok = T && img && hipMalloc(&d_tab, sizeof(tab)) == hipSuccess && hipMalloc(&d_count, sizeof(count)) == hipSuccess && hipMalloc(&d_bad, sizeof(bad)) == hipSuccess;
    ok = ok && hipMemsetAsync(d_tab, 0xff, sizeof(tab), st) == hipSuccess && hipMemsetAsync(d_count, 0, sizeof(count), st) == hipSuccess && hipMemsetAsync(d_bad, 0, sizeof(bad), st) == hipSuccess;
    if (ok) { blockrow_keys<<<1024, 256, 0, st>>>(nbr, b, p->rowpat, d_tab, d_count, d_bad); ok = hipGetLastError() == hipSuccess; }
    if (ok) { csr_max_column<<<2048, 256, 0, st>>>(p->n, ptr, p->codes, p->dict, d_bad + 1); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(tab, d_tab, sizeof(tab), hipMemcpyDeviceToHost, st) == hipSuccess && hipMemcpyAsync(count, d_count, sizeof(count), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(T, p->ptab, sizeof(int) * (size_t)p->ptab_len, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(img, p->vrecw, obytes + vbytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (d_tab) (void)hipFree(d_tab);
    if (d_count) (void)hipFree(d_count);
    if (d_bad) (void)hipFree(d_bad);
    const double *vals = reinterpret_cast<const double *>(img + obytes);
    // the rows of the block row named by `key`: their patterns, the common length, the columns relative to the block row's first row; false when they do not share them
    auto block_row = [&](int key, int *pats, int *offs) -> int {
        pats[0] = key;
        for (int k = 1; k < b; k++) { const unsigned int v = tab[key * 3 + k - 1]; if (v >= (unsigned int)NP) return 0; pats[k] = (int)v; }
        const int l = T[key + 1] - T[key];
        if (l < 1 || l > PATW_LEN) return 0;
        for (int j = 0; j < l; j++) offs[j] = T[NP + 1 + T[key] + j];
        for (int k = 1; k < b; k++) {
            if (T[pats[k] + 1] - T[pats[k]] != l) return 0;
            for (int j = 0; j < l; j++) if (T[NP + 1 + T[pats[k]] + j] + k != offs[j]) return 0;
        }
        return l;
    };
    while (ok && bad[0] == 0 && bad[1] >= 1) {                  // (one pass; `break` = does not qualify)
        int dom = 0;
        for (int i = 1; i < NP; i++) if (count[i] > count[dom]) dom = i;
        if (count[dom] * 2 < (unsigned long long)nbr) break;
        int dpat[4], doff[PATW_LEN];
        const int l = block_row(dom, dpat, doff);
        if (l < 1) break;
        int offs[PATW_LEN], start[16], mlen[16], base[17], nruns = 0;
        for (int j = 0; j < l; j++) offs[j] = doff[j];
        for (int a = 1; a < l; a++) { const int v = offs[a]; int c = a - 1; while (c >= 0 && offs[c] > v) { offs[c + 1] = offs[c]; c--; } offs[c + 1] = v; }
        bool fits = true;
        for (int j = 0; j < l && fits; ) {
            int e = j + 1;
            while (e < l && offs[e] == offs[e - 1] + 1) e++;
            if ((e < l && offs[e] == offs[e - 1]) || nruns == 16) { fits = false; break; }
            start[nruns] = offs[j]; mlen[nruns++] = e - j;
            j = e;
        }
        if (!fits) break;
        base[0] = 0;
        for (int a = 0; a < nruns; a++) base[a + 1] = base[a] + (((WAVE - 1) * b + mlen[a] + 1) & ~1);      // 64 block rows need 63 b + m columns of a run of m; even widths (pairs of slots)
        const int slots = base[nruns], nl = (slots + 2 * WAVE - 1) / (2 * WAVE);
        if (nl > (b == 2 ? 6 : b == 3 ? 8 : 12)) break;
        BlockDom D;
        memset(&D, 0, sizeof(D));
        D.len = l; D.key = dom; D.slots = slots; D.maxcol = bad[1]; D.b = b;
        for (int j = 0; j < l; j++) {
            int q = 0;
            for (int a = 0; a < nruns; a++) if (doff[j] >= start[a] && doff[j] < start[a] + mlen[a]) q = a;
            D.slot[j] = base[q] + (doff[j] - start[q]);
            for (int k = 0; k < b; k++) D.val[k][j] = vals[(size_t)dpat[k] * PATW_LEN + j];
        }
        D.pair = (l % 2 == 0) ? 1 : 0;
        for (int q = 0; q < l / 2 && D.pair; q++) if (D.slot[2 * q + 1] != D.slot[2 * q] + 1) D.pair = 0;
        const int nlk = b == 2 ? (nl <= 4 ? 4 : 6) : b == 3 ? (nl <= 6 ? 6 : 8) : (nl <= 8 ? 8 : 12);      // the kernel's instantiations
        int *stage = (int *)calloc((size_t)nlk * WAVE, sizeof(int));
        unsigned long long *rec = (unsigned long long *)calloc(256, sizeof(unsigned long long));
        if (stage && rec) {
            for (int k = 0; k < nlk; k++)
                for (int lane = 0; lane < WAVE; lane++) {
                    const int sl = 2 * (k * WAVE + lane);
                    int q = -1;
                    for (int a = 0; a < nruns; a++) if (sl >= base[a] && sl < base[a + 1]) q = a;
                    stage[k * WAVE + lane] = q >= 0 ? start[q] + (sl - base[q]) : 0;
                }
            for (int i = 0; i < 256; i++) {
                rec[i] = 1ull << 32;                              // foreign unless shown otherwise
                if (i >= NP || count[i] == 0) continue;
                int qpat[4], qoff[PATW_LEN];
                const int li = block_row(i, qpat, qoff);
                if (li < 1) continue;
                unsigned long long bits = 0;
                int j = 0;
                for (int sl = 0; sl < l && j < li; sl++)
                    if (qoff[j] == doff[sl]) {
                        bool same = true;
                        for (int k = 0; k < b; k++) same = same && memcmp(&vals[(size_t)qpat[k] * PATW_LEN + j], &D.val[k][sl], 8) == 0;
                        if (!same) break;
                        bits |= 1ull << sl; j++;
                    }
                if (j == li) rec[i] = bits;
            }
            if (hipMalloc(&p->bdrec, sizeof(unsigned long long) * 256) == hipSuccess && hipMalloc(&p->bstage, sizeof(int) * (size_t)nlk * WAVE) == hipSuccess &&
                hipMemcpy(p->bdrec, rec, sizeof(unsigned long long) * 256, hipMemcpyHostToDevice) == hipSuccess &&
                hipMemcpy(p->bstage, stage, sizeof(int) * (size_t)nlk * WAVE, hipMemcpyHostToDevice) == hipSuccess) { p->bd = D; if (b == 2) try_block2_march(p, doff, D, st); }
            else {
                if (p->bdrec) { (void)hipFree(p->bdrec); p->bdrec = nullptr; }
                if (p->bstage) { (void)hipFree(p->bstage); p->bstage = nullptr; }
            }
        }
        free(stage); free(rec);
        break;
    }
    free(T); free(img);
    return 0;
}
// b when the plan keeps block rows for spmv_csr_blockrows_staged_kernel, else 0
extern "C" int liship_csr_plan_block_rows(liship_csr_plan_t p) { return (p && p->vrecw && p->bdrec && p->bstage && p->bd.len > 0) ? p->bd.b : 0; }
// the planes of a 7-point grid in which the marching kernel needs neither pattern bytes nor masks (its BOX form), 0 when there are none
extern "C" int liship_csr_plan_box_planes(liship_csr_plan_t p) { return (p && p->drec && p->vrec) ? p->box_z1 - p->box_z0 : 0; }
// 1 when the plan also names a dominant pattern (spmv_csr_valuerec_dom_kernel), else 0
extern "C" int liship_csr_plan_dominant_pattern(liship_csr_plan_t p) { return (p && p->ptab8 && p->drec) ? (p->vrec ? 1 : 2) : 0; }      // 2: offsets only (no value records)

// Block-local columns for the products kernel (see spmv_csr_local_kernel): setup-time, optional, never an error when the
// matrix does not qualify.  Kept when the lists cover >= 90 % of the non-zeros and hold at most half as many columns as
// the blocks hold entries -- below that the 2 B + 4 B of a once-used column cost more than its 4 B index.
// Lists made of triples (3 unknowns per node: the sorted distinct columns of a row block are 3c, 3c + 1, 3c + 2 for the nodes it touches).  One pass over the lists:
// a block's true length (the padding repeats its last entry), whether it is a multiple of three and every group of three consecutive; when ALL blocks pass, the
// triples' first columns are kept beside the lists (a third of their size) and the kernel reads those.  Optional: any failure leaves the plan as it was.
namespace {
__global__ void local_runs_count(int nb, const int *__restrict__ doff, const int *__restrict__ dcol, int *__restrict__ nruns)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int d0 = doff[b], d1 = doff[b + 1];
    int nd = d1 - d0;
    while (nd > 1 && dcol[d0 + nd - 1] == dcol[d0 + nd - 2]) nd--;      // (strictly increasing up to the padding)
    bool ok = nd % 3 == 0;
    for (int k = 0; ok && k < nd; k += 3) ok = dcol[d0 + k + 1] == dcol[d0 + k] + 1 && dcol[d0 + k + 2] == dcol[d0 + k] + 2;
    nruns[b] = ok ? nd / 3 : -1;
}
__global__ void local_runs_write(int nb, const int *__restrict__ doff, const int *__restrict__ dcol, const int *__restrict__ droff, int *__restrict__ drun)
{
    const int b = blockIdx.x;
    const int d0 = doff[b], q0 = droff[b], nr = droff[b + 1] - q0;
    for (int j = threadIdx.x; j < nr; j += blockDim.x) drun[q0 + j] = dcol[d0 + 3 * j];
}
}
static void build_local_runs(liship_csr_plan_s *p, hipStream_t st)
{
    const int nb = p->nblocks;
    if (nb <= 0 || !p->dcol || !p->doff) return;
    int *d_n = nullptr, *h = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
    bool ok = h && hipMalloc(&d_n, sizeof(int) * (size_t)(nb + 1)) == hipSuccess;
    if (ok) { local_runs_count<<<(nb + 255) / 256, 256, 0, st>>>(nb, p->doff, p->dcol, d_n); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(h, d_n, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    long long total = 0;
    if (ok) {
        for (int b = 0; b < nb && ok; b++) { if (h[b] < 0) ok = false; else { const int c = h[b]; h[b] = (int)total; total += c; } }
        h[nb] = (int)total;
    }
    ok = ok && total > 0 && total < 0x7fffffffLL;
    if (ok) ok = hipMemcpyAsync(d_n, h, sizeof(int) * (size_t)(nb + 1), hipMemcpyHostToDevice, st) == hipSuccess && hipMalloc(&p->drun, sizeof(int) * (size_t)(total + 4)) == hipSuccess;
    if (ok) { local_runs_write<<<nb, 256, 0, st>>>(nb, p->doff, p->dcol, d_n, p->drun); ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess; }
    free(h);
    if (ok) p->droff = d_n;
    else {
        (void)hipGetLastError();
        if (d_n) (void)hipFree(d_n);
        if (p->drun) { (void)hipFree(p->drun); p->drun = nullptr; }
    }
}
extern "C" int liship_csr_plan_local_runs(liship_csr_plan_t p) { return (p && p->lcol && p->drun && p->droff) ? 3 : 0; }
static int g_local_runs = 1;
extern "C" int liship_spmv_csr_set_local_runs(int on) { g_local_runs = on ? 1 : 0; return 0; }
extern "C" int liship_spmv_csr_set_local_pairs(int on) { g_local_pairs = on ? 1 : 0; return 0; }
// 0: plans of short rows (mean < 22 entries) never try block-local columns (rounds 2-5; A/B).  Plans built from now on.
extern "C" int liship_spmv_csr_set_local_short_rows(int on) { g_local_short_rows = on ? 1 : 0; return 0; }

extern "C" int liship_csr_plan_localize_columns(liship_csr_plan_t p, const int *ptr, const int *idx, void *stream)
{
    if (!p || (p->n > 0 && (!ptr || !idx))) return LISHIP_ERR_ARG;
    if (p->lcol || p->codes || p->nblocks <= 0 || p->nnz <= 0 || g_variant != 0 || !aligned16(idx)) return 0;
    // Short rows too (round 6): the lists are TRIED for every plan without column codes whose rows hold 4 entries or more on average, and kept by the same rule --
    // the listed blocks cover 90 % of the entries with at most two listed columns per three entries.  An unstructured mesh with one unknown per node (ragged rows of
    // 8 .. 40 entries, numbered along a space-filling curve) qualifies, and the block-local kernel -- lanes own entries, x staged once per distinct column -- beats
    // the row-gather kernel's lane-per-row there at every mean row length measured (2 M nodes: mean 4.9 +6 %, 7.9 +15 %, 11 +23 %, 14.4 +36 %, 18.3 +39 %): ragged
    // rows leave a lane-per-row wavefront waiting for its longest row, and every gather of a lane-per-row wavefront touches 64 lines.  A plan that does not qualify
    // goes back to the row-gather kernel exactly as it was.
    const bool trial = !p->products;
    if (trial && (!g_local_short_rows || (double)p->nnz < 4.0 * p->n)) return 0;
    hipStream_t st = as_stream(stream);
    constexpr Geometry g = kGeom[LOCAL_GEOM];
    constexpr int NDMAX = 4 * g.block;      // up to four distinct columns per lane (the kernel's NDPL = 4 form)
    const int geom_before = p->geom;
    if (trial) p->products = 1;
    auto give_up = [&]() -> int {           // the plan as it was: the row-gather kernel's split for a short-row plan, the products kernel's own otherwise
        if (trial) { p->products = 0; p->local_trial_failed = 1; }
        if (geom_before != p->geom) { p->geom = geom_before; return build_split(p, ptr, st); }
        return trial ? build_split(p, ptr, st) : 0;
    };
    int *nd_dev = nullptr, *off = nullptr;
    int nb = 0, capl = 0;
    // blocks of 3584 items with the positions in registers (round 4); when those list more than two columns per lane, blocks of 3072.
    // (liship_spmv_csr_set_local_register_positions(0): the round-3 form, 4096-item blocks whatever the lists)
    for (int attempt = 0; attempt < 2; attempt++) {
        const int want = !g_local_rpos ? LOCAL_GEOM : attempt == 0 ? LOCAL_GEOM_R : LOCAL_GEOM4;
        if (p->geom != want) { p->geom = want; const int rc = build_split(p, ptr, st); if (rc) return rc; }
        nb = p->nblocks;
        capl = kGeom[want].work + SLACK;
        if (nd_dev) (void)hipFree(nd_dev);
        free(off);
        nd_dev = nullptr; off = nullptr;
        HIP_TRY(hipMalloc(&nd_dev, sizeof(int) * (size_t)(nb + 1)));
        if (capl <= 3840) csr_local_build<256, 0, 4096, 3840><<<nb, 256, 0, st>>>(p->blk, idx, capl, NDMAX, nd_dev, nullptr, nullptr, nullptr);
        else csr_local_build<256, 0, 8192><<<nb, 256, 0, st>>>(p->blk, idx, capl, NDMAX, nd_dev, nullptr, nullptr, nullptr);
        hipError_t e0 = hipGetLastError();
        off = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
        if (!off && e0 == hipSuccess) e0 = hipErrorOutOfMemory;
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(off, nd_dev, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, st);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) {
            (void)hipFree(nd_dev); free(off);
            (void)give_up();
            return (int)e0;
        }
        int most = 0;
        for (int b = 0; b < nb; b++) if (off[b] > most) most = off[b];
        if (!g_local_rpos || attempt == 1 || most <= 2 * g.block) break;      // short lists keep the larger blocks
    }
    hipError_t e = hipSuccess;
    long long listed = 0, covered = 0, run = 0;
    int ndmost = 0;
    for (int b = 0; b < nb; b++) {
        const int nd = off[b];
        if (nd > ndmost) ndmost = nd;
        off[b] = (int)run;
        if (nd > 0) { listed += nd; covered += p->blk_host[b + 1].y - p->blk_host[b].y; run += (nd + 3) & ~3; }
    }
    off[nb] = (int)run;
    // kept when the listed blocks cover 90 % of the entries with at most one listed column per two entries (long rows) / two per three (short rows: measured
    // on the 8 M-node mesh at 0.57 listed columns per entry, 0.574 -> 0.462 ms against the row-gather kernel)
    if (run > 0x7fffffffLL || covered * 10 < p->nnz * 9 || (trial ? listed * 3 > covered * 2 : listed * 2 > covered)) {      // not worth it: back to the plan's own split
        (void)hipFree(nd_dev); free(off);
        return give_up();
    }
    const size_t lbytes = ((size_t)p->nnz + 7) / 8 * 16 + 16 * WAVE;          // whole 16 B pieces, one wave slice of slack
    e = hipMalloc(&p->dcol, sizeof(int) * (size_t)(run + 4));
    if (e == hipSuccess) e = hipMalloc(&p->lcol, lbytes);
    if (e == hipSuccess) e = hipMemsetAsync(p->lcol, 0, lbytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nd_dev, off, sizeof(int) * (size_t)(nb + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        if (capl <= 3840) csr_local_build<256, 1, 4096, 3840><<<nb, 256, 0, st>>>(p->blk, idx, capl, NDMAX, nullptr, nd_dev, p->dcol, p->lcol);
        else csr_local_build<256, 1, 8192><<<nb, 256, 0, st>>>(p->blk, idx, capl, NDMAX, nullptr, nd_dev, p->dcol, p->lcol);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    free(off);
    if (e != hipSuccess) {
        if (p->dcol) (void)hipFree(p->dcol);
        if (p->lcol) (void)hipFree(p->lcol);
        (void)hipFree(nd_dev);
        p->dcol = nullptr; p->lcol = nullptr;
        (void)give_up();
        return (int)e;
    }
    p->doff = nd_dev;
    p->ndcol = run;
    build_local_runs(p, st);
    p->ndpl = ndmost > 2 * g.block ? 4 : 2;        // lists of up to 1024 columns (the dofs-per-node patterns in mesh order): two per lane, 8 KB of LDS; longer ones four
    p->xcap = ndmost <= 1024 ? 1024 : (ndmost <= 1536 && p->geom == LOCAL_GEOM4) ? 1536 : 2048;
    return 0;
}
#include "csr_order.hpp"
// ---------------------------------------------------------------------------------------------- reordering (round 5)
// A mesh whose nodes are numbered without locality (the Queen-class stand-in: numbers permuted at random inside runs of 1024 nodes) gives every row block ~2.3 x the
// distinct columns the same mesh has in a local numbering, each on a cache line of its own: the block-local kernel then spends more L1 <-> L2 requests on x than on the
// matrix (profiles/r04_queen_class_pmc.txt: 30 M of 55.7 M per product) and runs at 0.65 ms where the naturally numbered mesh takes 0.52.  The product does not care in
// which order rows are WALKED or what a column is CALLED -- a row sum is its own terms in their stored order -- so the plan may renumber: rounds 4-5 by a Cuthill-McKee
// walk of the matrix graph on the host (1.4-1.6 s there), round 6 by landmark distances on the device (csr_order.hpp: breadth-first searches, Morton keys, a stable
// radix sort -- the same locality, no host walk, index[] never leaves HBM), P A P^T built once in HBM.
namespace {
__global__ void csr_reorder_inverse(int n, const int *__restrict__ perm, int *__restrict__ inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}
// one wavefront per row of P A P^T: entry j of new row r is entry j of row perm[r], its column renumbered
__global__ __launch_bounds__(256)
void csr_reorder_rows(int n, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val, const int *__restrict__ perm,
                      const int *__restrict__ inv, const int *__restrict__ ptr2, int *__restrict__ idx2, double *__restrict__ val2, int *__restrict__ bad, int ncols)
{
    const int r = blockIdx.x * (256 / WAVE) + (int)threadIdx.x / WAVE, lane = (int)threadIdx.x & (WAVE - 1);
    if (r >= n) return;
    const int src = perm[r], s = ptr[src], len = ptr[src + 1] - s, d = ptr2[r];
    for (int j = lane; j < len; j += WAVE) {
        const int c = idx[s + j];
        if (c >= n && c < ncols) idx2[d + j] = c;                  // a rank's ghost column: not renumbered (the halo lands where it always did)
        else if (c < 0 || c >= n) { *bad = 1; idx2[d + j] = 0; }   // a column outside the matrix
        else idx2[d + j] = inv[c];
        val2[d + j] = val[s + j];
    }
}

} // namespace

// 128 B lines of x (columns >> 4) the row blocks of a plan touch, summed over the blocks; -1: could not be counted
static long long block_lines(const liship_csr_plan_s *p, const int *idx, hipStream_t st)
{
    const int nb = p->nblocks;
    if (nb <= 0) return 0;
    int *nd_dev = nullptr;
    if (hipMalloc(&nd_dev, sizeof(int) * (size_t)(nb + 1)) != hipSuccess) { (void)hipGetLastError(); return -1; }
    std::vector<int> nd((size_t)nb);
    if (kGeom[p->geom].work + SLACK <= 3840) csr_local_build<256, 0, 4096, 3840><<<nb, 256, 0, st>>>(p->blk, idx, kGeom[p->geom].work + SLACK, 0x7fffffff, nd_dev, nullptr, nullptr, nullptr, 4);
    else csr_local_build<256, 0, 8192><<<nb, 256, 0, st>>>(p->blk, idx, kGeom[p->geom].work + SLACK, 0x7fffffff, nd_dev, nullptr, nullptr, nullptr, 4);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(nd.data(), nd_dev, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(nd_dev);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    long long total = 0;
    for (int b = 0; b < nb; b++) total += nd[b];
    return total;
}

extern "C" int liship_spmv_csr_set_reorder(int mode) { g_reorder = mode == 2 ? 2 : mode ? 1 : 0; return 0; }
// listed columns of the reordered form (compare liship_csr_plan_localized: the original numbering's), 0 when the plan has none
extern "C" long long liship_csr_plan_reordered(liship_csr_plan_t p) { return (p && p->inner) ? p->inner->ndcol : 0; }

// Builds the reordered form when the plan keeps block-local columns AND its lists are long (more than one listed column per `min_items_per_listed` non-zeros: 4 by
// default when 0 is passed) AND the renumbered matrix lists at most 3/4 of them; for short rows (the row-gather kernel) the 128 B lines of x a row block touches take
// the lists' place (more than one per 4 entries; at most half of them afterwards).  Never an error when the matrix does not qualify; out of memory (2) leaves the plan
// as it was.  The numbering is found in HBM (csr_order.hpp); the host sees ptr[] (4 B per row) and the permutation.
static int reorder_impl(liship_csr_plan_t p, const int *ptr, const int *idx, const double *val, int min_items_per_listed, const int *hint, void *stream);
extern "C" int liship_csr_plan_reorder(liship_csr_plan_t p, const int *ptr, const int *idx, const double *val, int min_items_per_listed, void *stream)
{
    return reorder_impl(p, ptr, idx, val, min_items_per_listed, nullptr, stream);
}
// the same with a permutation to try first (host, n entries: new position -> row; e.g. the one a plan of the same sparsity pattern found -- a matrix whose values were
// edited needs a new plan but not a new walk).  Anything that is not a permutation of 0 .. n-1, or that does not shorten the lists enough, is dropped for a walk.
extern "C" int liship_csr_plan_reorder_with(liship_csr_plan_t p, const int *ptr, const int *idx, const double *val, int min_items_per_listed, const int *perm_hint, void *stream)
{
    if (perm_hint) {
        const int rc = reorder_impl(p, ptr, idx, val, min_items_per_listed, perm_hint, stream);
        if (rc || (p && p->inner)) return rc;
    }
    return reorder_impl(p, ptr, idx, val, min_items_per_listed, nullptr, stream);
}
// A rank's local matrix in a multi-rank job: columns [n, ncols) are ghost columns (x[n .. ncols) is filled by the halo exchange).  Said BEFORE liship_csr_plan_reorder:
// the numbering is then found on the owned columns alone (the ghosts are no vertices of the local graph), the ghost columns keep their numbers in P A P^T, and the rows
// that read one are placed behind all the others, so that rows [0, liship_csr_plan_reordered_inner_rows) can run while the halo travels.
extern "C" int liship_csr_plan_set_ghost_columns(liship_csr_plan_t p, int ncols)
{
    if (!p || ncols < p->n) return LISHIP_ERR_ARG;
    p->ncols = ncols;
    return 0;
}
// 1: the plan has NO block-local lists although it wanted them (a short-row plan whose trial failed; a long-row plan whose row blocks list too many columns): the
// numbering has no locality at all, the product runs at 30-40 % of its roofline and a renumbered form doubles it -- the host layer builds that form early for such a plan
extern "C" int liship_csr_plan_lists_failed(liship_csr_plan_t p) { return (p && !p->lcol && !p->codes && (p->local_trial_failed || p->products)) ? 1 : 0; }
extern "C" int liship_csr_plan_reordered_inner_rows(liship_csr_plan_t p) { return (p && p->inner) ? (p->ncols > p->n ? p->r_inner_end : p->n) : 0; }
// the permutation of the reordered form to the host (n entries); LISHIP_ERR_ARG when the plan has none
extern "C" int liship_csr_plan_reorder_permutation(liship_csr_plan_t p, int *out_host)
{
    if (!p || !p->inner || !p->r_perm || !out_host) return LISHIP_ERR_ARG;
    HIP_TRY(hipMemcpy(out_host, p->r_perm, sizeof(int) * (size_t)p->n, hipMemcpyDeviceToHost));
    return 0;
}
static int reorder_impl(liship_csr_plan_t p, const int *ptr, const int *idx, const double *val, int min_items_per_listed, const int *hint, void *stream)
{
    if (!p || (p->n > 0 && (!ptr || !idx || !val))) return LISHIP_ERR_ARG;
    if (p->inner || p->codes || p->rowpat || p->n < 65536 || p->nnz <= 0 || g_variant != 0) return 0;
    const int mi = min_items_per_listed > 0 ? min_items_per_listed : 4;
    if (p->lcol && p->ndcol * (long long)mi <= p->nnz) return 0;   // lists short already: the numbering is local (no lists at all: too many distinct columns per row block)
    hipStream_t st = as_stream(stream);
    const int n = p->n, ncols = p->ncols > p->n ? p->ncols : p->n;
    const size_t nnz = (size_t)p->nnz;
    int inner_rows = n;
    // short rows (the row-gather kernel: no lists to judge by): the 128 B lines of x a row block touches.  A grid in its natural order touches ~0.05 per entry (the
    // same lines serve a block's neighbouring rows), a numbering without locality ~1
    long long lines_before = 0;
    if (!p->products) {
        lines_before = block_lines(p, idx, st);
        if (lines_before < 0) return 0;                            // (could not count: leave the plan alone)
        // (round 6: ... unless the plan tried block-local columns and its lists were too long -- few lines per row block and yet a column of its own for nearly every entry is
        //  what a mesh numbered at random inside coarse cells looks like: the row-gather kernel's 64 lanes then gather from 64 different lines at every step, 42 % of the
        //  roofline at 8 M nodes, and the same matrix renumbered runs the block-local kernel at 70 %+)
        if (lines_before * mi <= p->nnz && !p->local_trial_failed) return 0;
    }
    int *hptr = (int *)malloc(sizeof(int) * ((size_t)n + 1)), *hidx = nullptr, *order = (int *)malloc(sizeof(int) * (size_t)n);
    int *hptr2 = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    int *inv = nullptr;
    liship_csr_plan_s *in = nullptr;
    hipError_t e = (hptr && order && hptr2) ? hipSuccess : hipErrorOutOfMemory;
    bool keep = false, have_order = false;
    if (e == hipSuccess) e = hipMemcpyAsync(hptr, ptr, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, st);      // (row starts: 4 B per row; index[] stays in HBM)
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && hint) {                                 // a permutation of 0 .. n-1, or nothing
        std::vector<unsigned char> seen((size_t)n, 0);
        have_order = true;
        for (int i = 0; i < n && have_order; i++) {
            const int r = hint[i];
            if (r < 0 || r >= n || seen[r]) have_order = false; else { seen[r] = 1; order[i] = r; }
        }
    } else if (e == hipSuccess) have_order = order_dev::device_order(n, ncols, ptr, idx, (p->products || (double)p->nnz >= 12.0 * n) ? 6 : 3, order, &inner_rows, st);      // dense neighbourhoods (long rows; 12 entries per row or more: a k-nearest-neighbour mesh, 27-point connectivity): six landmarks; sparse ones (grid graphs, whose breadth-first distances ARE coordinates): three
    if (e == hipSuccess && have_order) {
        bool moved = false;
        hptr2[0] = 0;
        for (int r = 0; r < n; r++) { hptr2[r + 1] = hptr2[r] + (hptr[order[r] + 1] - hptr[order[r]]); moved = moved || order[r] != r; }
        if (moved) {
            e = hipMalloc(&p->r_perm, sizeof(int) * ((size_t)n + 2));
            if (e == hipSuccess) e = hipMalloc(&p->r_ptr, sizeof(int) * ((size_t)n + 2));
            if (e == hipSuccess) e = hipMalloc(&inv, sizeof(int) * ((size_t)n + 4));
            if (e == hipSuccess) e = hipMemsetAsync(inv + n, 0, sizeof(int), st);            // inv[n]: the kernel's flag
            if (e == hipSuccess) e = hipMalloc(&p->r_idx, sizeof(int) * (nnz + 16));
            if (e == hipSuccess) e = hipMalloc(&p->r_val, sizeof(double) * (nnz + 16));
            if (e == hipSuccess) e = hipMalloc(&p->r_x, sizeof(double) * ((size_t)n + 2));
            if (e == hipSuccess) e = hipMemcpyAsync(p->r_perm, order, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(p->r_ptr, hptr2, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemsetAsync(p->r_idx + nnz, 0, sizeof(int) * 16, st);
            if (e == hipSuccess) e = hipMemsetAsync(p->r_val + nnz, 0, sizeof(double) * 16, st);
            if (e == hipSuccess) {
                csr_reorder_inverse<<<(n + 255) / 256, 256, 0, st>>>(n, p->r_perm, inv);
                csr_reorder_rows<<<(n + 3) / 4, 256, 0, st>>>(n, ptr, idx, val, p->r_perm, inv, p->r_ptr, p->r_idx, p->r_val, inv + n, ncols);
                e = hipGetLastError();
            }
            int bad = 0;
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, inv + n, sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess && !bad) {
                int rc = liship_csr_plan_create(&in, n, p->r_ptr, stream);
                if (!rc) rc = liship_csr_plan_localize_columns(in, p->r_ptr, p->r_idx, stream);      // (short rows too, round 6: a renumbered short-row matrix whose row blocks now share their columns takes the block-local kernel)
                if (rc) e = (hipError_t)rc;
                else if (in->products) keep = in->lcol && (!p->lcol || in->ndcol * 4 <= p->ndcol * 3);
                else {                                              // short rows: at most half the lines
                    const long long lines_after = block_lines(in, p->r_idx, st);
                    keep = lines_after >= 0 && lines_after * 2 <= lines_before;
                    if (keep) in->ndcol = lines_after;              // (what liship_csr_plan_reordered reports for such a plan)
                }
            }
        }
    }
    if (getenv("LIS_AMD_REORDER_TRACE"))            // why a matrix did (not) get a renumbered form
        fprintf(stderr, "liblis_amd: reorder: n %d nnz %lld long rows %d lists before %lld lines before %lld -> ordering %s, candidate plan %s, its lists %lld: %s (HIP %d)\n", n, p->nnz, p->products,
                p->lcol ? p->ndcol : 0ll, lines_before, have_order ? "found" : "none", in ? (in->lcol ? "block-local" : "row-gather") : "none", in ? in->ndcol : 0ll, keep ? "kept" : "dropped", (int)e);
    free(hptr); free(hidx); free(order); free(hptr2);
    if (inv) (void)hipFree(inv);
    if (keep) { in->first_term = p->first_term; in->ncols = p->ncols; p->inner = in; p->r_inner_end = (ncols > n && !hint) ? inner_rows : 0; return 0; }
    if (in) (void)liship_csr_plan_destroy(in);
    if (p->r_ptr) (void)hipFree(p->r_ptr);
    if (p->r_idx) (void)hipFree(p->r_idx);
    if (p->r_perm) (void)hipFree(p->r_perm);
    if (p->r_val) (void)hipFree(p->r_val);
    if (p->r_x) (void)hipFree(p->r_x);
    p->r_ptr = p->r_idx = p->r_perm = nullptr; p->r_val = p->r_x = nullptr;
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// entries of the distinct-column lists when the plan keeps block-local columns, 0 otherwise
extern "C" long long liship_csr_plan_localized(liship_csr_plan_t p) { return (p && p->lcol) ? p->ndcol : 0; }
extern "C" int liship_spmv_csr_set_local_columns(int on) { g_local_cols = on ? 1 : 0; return 0; }
// 0: plans built from now on take the round-3 form of the block-local kernel (4096-item blocks, positions staged in LDS: 3 / 2 workgroups per CU); A/B, same bits
// 0: the 7-offset pattern kernel walks the row blocks in their natural order (round-robin over the XCDs) instead of XCD strips; A/B measurements, same bits
extern "C" int liship_spmv_csr_set_xcd_strips(int on) { g_xcd_strips = on ? 1 : 0; return 0; }

// The band of a matrix whose plan has no row patterns (4 B indices, one-byte codes): the largest |column - row| among the owned columns, and how many rows reach
// it.  When most rows do -- the +-plane neighbours of a 3-D grid -- that distance is the plane the XCD strips are cut from (xcd_strip_unit), exactly what the
// largest pattern offset is for patterned plans.  Two passes over index[] at plan time; an order of the row blocks only: the bits cannot depend on it.
namespace {
__global__ void csr_band_max(int n, const int *__restrict__ ptr, const int *__restrict__ idx, int *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (r < n) for (int k = ptr[r], e = ptr[r + 1]; k < e; k++) { const int c = idx[k]; if (c < n) m = max(m, abs(c - r)); }
    for (int s = WAVE / 2; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}
__global__ void csr_band_count(int n, const int *__restrict__ ptr, const int *__restrict__ idx, const int *__restrict__ band, unsigned long long *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, B = band[0];
    bool hit = false;
    if (r < n) for (int k = ptr[r], e = ptr[r + 1]; k < e; k++) { const int c = idx[k]; hit = hit || (c < n && abs(c - r) == B); }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(hit);
    if ((threadIdx.x & (WAVE - 1)) == 0 && b) atomicAdd(out, (unsigned long long)__builtin_popcountll(b));
}
}
extern "C" int liship_csr_plan_scan_band(liship_csr_plan_t p, const int *ptr, const int *idx, void *stream)
{
    if (!p || !ptr || !idx) return LISHIP_ERR_ARG;
    if (p->xs_rows > 0 || p->products || p->n < (1 << 16) || p->nnz <= 0) return 0;      // the patterns named the plane / another kernel family / too small to matter
    hipStream_t st = as_stream(stream);
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), st);
    const int threads = 256, grid = (p->n + threads - 1) / threads;
    if (e == hipSuccess) { csr_band_max<<<grid, threads, 0, st>>>(p->n, ptr, idx, reinterpret_cast<int *>(d)); e = hipGetLastError(); }
    if (e == hipSuccess) { csr_band_count<<<grid, threads, 0, st>>>(p->n, ptr, idx, reinterpret_cast<const int *>(d), d + 1); e = hipGetLastError(); }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    const int band = (int)(h[0] & 0xffffffffull);
    if (band > 0 && 2 * h[1] >= (unsigned long long)p->n) p->xs_rows = band;
    return 0;
}
extern "C" int liship_csr_plan_strip_rows(liship_csr_plan_t p) { return p ? p->xs_rows : 0; }
extern "C" int liship_spmv_csr_set_local_register_positions(int on) { g_local_rpos = on ? 1 : 0; return 0; }
extern "C" int liship_spmv_csr_set_team(int on) { g_team = on ? 1 : 0; return 0; }
extern "C" int liship_spmv_csr_set_dom_march(int on) { g_dom_march = on; return 0; }
extern "C" int liship_spmv_csr_set_block_rows(int on) { g_block_rows = on == 2 ? 2 : on ? 1 : 0; return 0; }
extern "C" int liship_spmv_csr_set_wide_union(int on) { g_wide_union = on == 2 ? 2 : on ? 1 : 0; return 0; }

