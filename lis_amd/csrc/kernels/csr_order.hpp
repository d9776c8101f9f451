// csr_order.hpp -- a locality-restoring numbering of a matrix graph, found ON THE DEVICE (round 6; included by csr_plan.hpp, same translation unit).
//
// Rounds 4-5 renumbered badly numbered matrices by a Cuthill-McKee walk on the host: 1.15 GB of index[] over PCIe and a sequential breadth-first walk with a sort per
// vertex, 1.4-1.6 s on the Queen-class matrix ("a CDNA4-first library should not have a sequential host graph walk").  What the product kernels need from a
// numbering is not a small bandwidth but LOCALITY: the rows of a row block should share their columns.  Graph distances give that without a serial walk:
//
//   1. level-synchronous breadth-first searches (one kernel launch per level, every vertex of the current level marks its unvisited neighbours: a benign race, all
//      writers store the same level) from L landmarks chosen far apart -- a = the vertex farthest from vertex 0, b = farthest from a, c = farthest from {a, b}, and so on
//      (ties: the smallest index), so the L distance fields are a coordinate system of the mesh;
//   2. key(v) = the bit-interleaved (Morton) code of those distances: vertices with close keys are close in the graph;
//   3. a stable LSD radix sort of (key, index) -- per-tile histograms, one scan, a stable scatter: deterministic, no atomics on the order -- gives the numbering.
//
// Measured on the CPU prototype (the scrambled 3-dof FEM mesh of the tests, listed columns per 4096-item row block, summed): caller's numbering 4.81 M, natural grid order
// 1.92 M, reverse Cuthill-McKee 1.91 M, THIS with 4 landmarks 1.88 M (3 landmarks: 2.45 M); a 7-point grid numbered at random, 128 B lines of x per 2048-item row block:
// 590 k -> 26 k with 3 landmarks (natural 24 k, RCM 23 k; 4 landmarks: 40 k).  At 527 k rows of the FEM mesh: caller's 13.4 M, natural 5.34 M, RCM 5.39 M, 3 / 4 / 5 / 6
// landmarks 7.02 / 5.83 / 4.67 / 4.21 M -- compact 3-D cells beat line-by-line orders.  So: 6 landmarks for long rows (dense, Chebyshev-like neighbourhoods), 3 for short ones.
// Components: up to MAX_COMP are ordered one after the other (each with its own landmarks); vertices without a neighbour and anything beyond keep their index order at the end.
// The result is only ever a CANDIDATE: reorder_impl keeps P A P^T when its lists / lines shrink enough, exactly as it judged the walk's.
#pragma once

namespace order_dev {

constexpr int MAX_COMP = 8, MAX_LEVEL = (1 << 15) - 1, LEVEL_BATCH = 32, SORT_TILE = 2048, MAX_FIELDS = 6;
struct Fields { const int *l[MAX_FIELDS]; };      // the landmark distance fields, by value into the kernels
constexpr int LEVEL_BUDGET = 16384;     // levels over ALL searches of one ordering: a graph of huge diameter (a chain) is not worth ~10 us per level -- no candidate then

// one level of a top-down breadth-first search: every vertex of level `cur` gives its unvisited neighbours level cur + 1
__global__ void bfs_level(int n, const int *__restrict__ ptr, const int *__restrict__ idx, int *__restrict__ level, int cur, int *__restrict__ grew)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || level[v] != cur) return;
    bool any = false;
    for (int k = ptr[v], e = ptr[v + 1]; k < e; k++) {
        const int c = idx[k];
        if (c >= 0 && c < n && c != v && level[c] < 0) { level[c] = cur + 1; any = true; }
    }
    if (any) grew[cur & (LEVEL_BATCH - 1)] = 1;
}
__global__ void fill_int(int n, int *__restrict__ a, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}
// vertices the last search from this component's seed reached and no earlier component owns: they are component `c`
__global__ void claim_component(int n, const int *__restrict__ level, int *__restrict__ comp, int c)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n && comp[v] < 0 && level[v] >= 0) comp[v] = c;
}
// the smallest index without a component that has a neighbour other than itself (an isolated vertex is nobody's seed): atomicMin into out[0]
__global__ void next_seed(int n, const int *__restrict__ ptr, const int *__restrict__ idx, const int *__restrict__ comp, int *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || comp[v] >= 0) return;
    bool nb = false;
    for (int k = ptr[v], e = ptr[v + 1]; k < e && !nb; k++) { const int c = idx[k]; nb = c >= 0 && c < n && c != v; }
    if (nb) atomicMin(out, v);
}
// the vertex of component `c` that maximises the smallest of the first `count` fields; ties go to the smallest index.  Packed into one 64-bit atomicMax.
__global__ void farthest(int n, const int *__restrict__ comp, int c, Fields F, int count, unsigned long long *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || comp[v] != c) return;
    int m = F.l[0][v];
    for (int k = 1; k < count; k++) m = min(m, F.l[k][v]);
    if (m < 0) return;
    atomicMax(out, ((unsigned long long)(unsigned)m << 32) | (unsigned long long)(0xffffffffu - (unsigned)v));
}
__device__ __forceinline__ unsigned long long spread(unsigned v, int fields, int bits)          // bit i of v -> bit fields * i
{
    unsigned long long r = 0;
    for (int b = 0; b < bits; b++) r |= (unsigned long long)((v >> b) & 1u) << (fields * b);
    return r;
}
// key = component (top byte) | Morton code of the landmark distances (each shifted down to `bits` bits); no component: the last bucket, in index order.
// ncols > n (a rank's local matrix): a row that reads a ghost column gets bit 62 -- all such rows sort behind the rows that read none (and in front of the last
// bucket); inner[0] counts the rows in front of them.
__global__ void make_keys(int n, int ncols, const int *__restrict__ ptr, const int *__restrict__ idx, int fields, int bits, int shift, const int *__restrict__ comp, Fields F,
                          unsigned long long *__restrict__ key, int *__restrict__ payload, int *__restrict__ inner)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    bool front = false;
    if (v < n) {
        payload[v] = v;
        const int c = comp[v];
        if (c < 0) key[v] = 0xffull << 56;
        else {
            unsigned long long k = 0;
            for (int f = 0; f < fields; f++) k |= spread((unsigned)min(max(F.l[f][v], 0) >> shift, (1 << bits) - 1), fields, bits) << f;
            bool ghost = false;
            if (ncols > n) for (int j = ptr[v], e = ptr[v + 1]; j < e && !ghost; j++) ghost = idx[j] >= n;
            front = !ghost;
            key[v] = ((unsigned long long)c << 56) | k | (ghost ? 1ull << 62 : 0ull);
        }
    }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(front);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(inner, __builtin_popcountll(b));
}

// ---- stable LSD radix sort, 8 bits a pass, (key, payload) pairs.  Tile = SORT_TILE consecutive elements per workgroup of 256 lanes.
__global__ __launch_bounds__(256) void radix_hist(int n, const unsigned long long *__restrict__ key, int shift, int ntiles, int *__restrict__ hist)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
    for (int i = threadIdx.x; i < SORT_TILE; i += 256) if (base + i < n) atomicAdd(&h[(key[base + i] >> shift) & 255], 1);
    __syncthreads();
    hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];          // digit-major: one exclusive scan over the whole array gives every (digit, tile) its start
}
// exclusive scan of `count` ints in place by ONE workgroup of 1024 lanes (count = 256 * tiles: half a million at most)
__global__ __launch_bounds__(1024) void scan_exclusive(int count, int *__restrict__ a)
{
    __shared__ int part[1024];
    const int per = (count + 1023) / 1024, b = threadIdx.x * per, e = min(count, b + per);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int t = 0; t < 1024; t++) { const int v = part[t]; part[t] = run; run += v; } }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = b; i < e; i++) { const int v = a[i]; a[i] = run; run += v; }
}
// stable scatter: the tile is walked in rounds of 256 consecutive elements; inside a round an element's rank among the equal digits before it is counted outright
__global__ __launch_bounds__(256) void radix_scatter(int n, const unsigned long long *__restrict__ key, const int *__restrict__ payload, int shift, int ntiles,
                                                     const int *__restrict__ start, unsigned long long *__restrict__ key_out, int *__restrict__ payload_out)
{
    __shared__ int next[256];
    __shared__ unsigned char dig[256];
    next[threadIdx.x] = start[threadIdx.x * ntiles + blockIdx.x];
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_TILE; r += 256) {
        const int i = base + r + (int)threadIdx.x;
        const bool live = i < n;
        unsigned long long k = 0;
        int d = 0;
        if (live) { k = key[i]; d = (int)((k >> shift) & 255); }
        dig[threadIdx.x] = (unsigned char)d;
        __syncthreads();
        const int lim = min(256, n - (base + r));                     // live elements of this round are its first `lim` lanes
        int rank = 0, same = 0;
        if (live) {
            for (int j = 0; j < lim; j++) { const bool eq = dig[j] == (unsigned char)d; rank += (eq && j < (int)threadIdx.x); same += eq; }
            const int at = next[d] + rank;
            key_out[at] = k; payload_out[at] = payload[i];
        }
        __syncthreads();
        if (live && rank == 0) next[d] += same;                        // one lane per digit present moves its cursor past the round
        __syncthreads();
    }
}

struct Scratch {
    int *level[1 + MAX_FIELDS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *comp = nullptr, *grew = nullptr, *hist = nullptr, *pay[2] = {nullptr, nullptr};
    unsigned long long *key[2] = {nullptr, nullptr}, *best = nullptr;
    ~Scratch()
    {
        for (int *p : level) if (p) (void)hipFree(p);
        for (int *p : pay) if (p) (void)hipFree(p);
        for (unsigned long long *p : key) if (p) (void)hipFree(p);
        if (comp) (void)hipFree(comp);
        if (grew) (void)hipFree(grew);
        if (hist) (void)hipFree(hist);
        if (best) (void)hipFree(best);
    }
};

// one breadth-first search from `src` into `level` (which holds -1 wherever this search may go); returns the number of levels, -1 on a runtime error
static int bfs(int n, const int *ptr, const int *idx, int *level, int src, int *grew, hipStream_t st, int *budget)
{
    const int grid = (n + 255) / 256, zero = 0;
    if (hipMemcpyAsync(level + src, &zero, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    int host[LEVEL_BATCH];
    for (int cur = 0; cur < MAX_LEVEL; cur += LEVEL_BATCH) {
        if ((*budget -= LEVEL_BATCH) < 0) return -1;
        if (hipMemsetAsync(grew, 0, sizeof(int) * LEVEL_BATCH, st) != hipSuccess) return -1;
        for (int l = 0; l < LEVEL_BATCH; l++) bfs_level<<<grid, 256, 0, st>>>(n, ptr, idx, level, cur + l, grew);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(host, grew, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        for (int l = 0; l < LEVEL_BATCH; l++) if (!host[l]) return cur + l + 1;       // level cur + l found nothing new: it was the last
    }
    return MAX_LEVEL;
}
static int pick_farthest(int n, const int *comp, int c, const Fields &F, int count, unsigned long long *best, hipStream_t st)
{
    unsigned long long h = 0;
    if (hipMemsetAsync(best, 0, sizeof(unsigned long long), st) != hipSuccess) return -1;
    farthest<<<(n + 255) / 256, 256, 0, st>>>(n, comp, c, F, count, best);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, best, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    if (h == 0) return -1;
    return (int)(0xffffffffu - (unsigned)(h & 0xffffffffull));
}

// order_host[new position] = row.  fields: how many landmarks (3 .. MAX_FIELDS).  false: a runtime error, out of memory or a graph of huge diameter (the plan then
// simply has no renumbered form)
static bool device_order(int n, int ncols, const int *ptr, const int *idx, int fields, int *order_host, int *inner_rows, hipStream_t st)
{
    if (n <= 0) return true;
    fields = fields < 3 ? 3 : fields > MAX_FIELDS ? MAX_FIELDS : fields;
    Scratch s;
    const size_t nb = sizeof(int) * ((size_t)n + 4);
    const int grid = (n + 255) / 256, ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    for (int k = 0; k <= fields; k++) if (hipMalloc(&s.level[k], nb) != hipSuccess) return false;
    if (hipMalloc(&s.comp, nb) != hipSuccess || hipMalloc(&s.grew, sizeof(int) * LEVEL_BATCH) != hipSuccess || hipMalloc(&s.best, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&s.hist, sizeof(int) * 256 * (size_t)ntiles) != hipSuccess) return false;
    for (int k = 0; k < 2; k++) if (hipMalloc(&s.key[k], sizeof(unsigned long long) * ((size_t)n + 2)) != hipSuccess || hipMalloc(&s.pay[k], nb) != hipSuccess) return false;
    for (int k = 0; k <= fields; k++) fill_int<<<grid, 256, 0, st>>>(n, s.level[k], -1);
    fill_int<<<grid, 256, 0, st>>>(n, s.comp, -1);
    int *l0 = s.level[0];
    Fields F{}, F0{};
    for (int f = 0; f < fields; f++) F.l[f] = s.level[1 + f];
    F0.l[0] = l0;
    int seed = 0, budget = LEVEL_BUDGET, deepest = 0;
    for (int c = 0; c < MAX_COMP; c++) {
        if (c > 0) {                                                    // the next component's seed: the smallest index nobody owns that has a neighbour
            int h = 0x7fffffff;
            if (hipMemcpyAsync(s.grew, &h, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return false;
            next_seed<<<grid, 256, 0, st>>>(n, ptr, idx, s.comp, s.grew);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, s.grew, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
            if (h == 0x7fffffff) break;
            seed = h;
        }
        if (bfs(n, ptr, idx, l0, seed, s.grew, st, &budget) < 0) return false;
        claim_component<<<grid, 256, 0, st>>>(n, l0, s.comp, c);
        // landmark f: the vertex of the component farthest from the seed (f = 0), then farthest from all the landmarks before it
        for (int f = 0; f < fields; f++) {
            const int at = f == 0 ? pick_farthest(n, s.comp, c, F0, 1, s.best, st) : pick_farthest(n, s.comp, c, F, f, s.best, st);
            if (at < 0) return false;
            const int depth = bfs(n, ptr, idx, s.level[1 + f], at, s.grew, st, &budget);
            if (depth < 0) return false;
            if (depth > deepest) deepest = depth;
        }
    }
    // 56 key bits for the Morton code: `bits` per field; deeper graphs lose their low distance bits (neighbouring levels share a cell)
    const int bits = 56 / fields > 15 ? 15 : 56 / fields;
    int shift = 0;
    while ((deepest >> shift) >= (1 << bits)) shift++;
    if (hipMemsetAsync(s.grew, 0, sizeof(int), st) != hipSuccess) return false;
    make_keys<<<grid, 256, 0, st>>>(n, ncols, ptr, idx, fields, bits, shift, s.comp, F, s.key[0], s.pay[0], s.grew);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(inner_rows, s.grew, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    int cur = 0;
    for (int sh = 0; sh < 64; sh += 8) {                                 // (passes over digits that are all zero cost a histogram and move nothing: kept simple)
        radix_hist<<<ntiles, 256, 0, st>>>(n, s.key[cur], sh, ntiles, s.hist);
        scan_exclusive<<<1, 1024, 0, st>>>(256 * ntiles, s.hist);
        radix_scatter<<<ntiles, 256, 0, st>>>(n, s.key[cur], s.pay[cur], sh, ntiles, s.hist, s.key[cur ^ 1], s.pay[cur ^ 1]);
        if (hipGetLastError() != hipSuccess) return false;
        cur ^= 1;
    }
    return hipMemcpyAsync(order_host, s.pay[cur], sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}

} // namespace order_dev
