"""ctypes wrapper around oracle/liblis_oracle.so (the CPU checker).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblis_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "liblis_ref.so")

I = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
ci, cd = C.c_int, C.c_double


class Result(C.Structure):
    _fields_ = [("iter", ci), ("retcode", ci), ("resid", cd)]


def build():
    """(Re)build the oracle and, when /root/reference is present, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "-j8"], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "lis_oracle.c")
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src)):
        build()
    L = C.CDLL(ORACLE_SO)
    P = C.c_void_p
    sig = {
        "orc_gen_poisson1d": (ci, [ci, ci, ci, I, I, D]),
        "orc_gen_poisson3d": (ci, [ci, ci, ci, ci, ci, ci, I, I, D]),
        "orc_spmv_csr": (None, [ci, I, I, D, D, D]),
        "orc_spmv_csc": (None, [ci, ci, I, I, D, D, D]),
        "orc_spmv_ell": (None, [ci, ci, I, D, D, D]),
        "orc_spmv_dia": (None, [ci, ci, ci, I, D, D, D]),
        "orc_spmv_jad": (None, [ci, ci, ci, I, I, I, D, D, D]),
        "orc_spmv_bsr": (None, [ci, ci, ci, ci, I, I, D, D, D]),
        "orc_spmvh_csr": (None, [ci, ci, I, I, D, D, D]),
        "orc_spmvh_csc": (None, [ci, I, I, D, D, D]),
        "orc_spmvh_ell": (None, [ci, ci, ci, I, D, D, D]),
        "orc_spmvh_dia": (None, [ci, ci, ci, I, D, D, D]),
        "orc_spmvh_jad": (None, [ci, ci, ci, I, I, I, D, D, D]),
        "orc_spmvh_bsr": (None, [ci, ci, ci, I, I, D, D, D, ci]),
        "orc_ell_maxnzr": (ci, [ci, I]),
        "orc_csr2ell": (None, [ci, I, I, D, ci, I, D]),
        "orc_csr2csc": (None, [ci, ci, I, I, D, I, I, D]),
        "orc_csr2dia": (ci, [ci, ci, I, I, D, P, P]),
        "orc_csr2jad": (None, [ci, I, I, D, ci, I, I, I, D]),
        "orc_csr2bsr": (ci, [ci, I, I, D, ci, ci, I, P, P]),
        "orc_csr_diagonal": (None, [ci, I, I, D, D]),
        "orc_dot": (cd, [ci, D, D]),
        "orc_nrm2": (cd, [ci, D]),
        "orc_nrm1": (cd, [ci, D]),
        "orc_axpy": (None, [ci, cd, D, D]),
        "orc_xpay": (None, [ci, D, cd, D]),
        "orc_axpyz": (None, [ci, cd, D, D, D]),
        "orc_scale": (None, [ci, cd, D]),
        "orc_pmul": (None, [ci, D, D, D]),
        "orc_reciprocal": (None, [ci, D]),
        "orc_cg": (Result, [ci, I, I, D, D, D, ci, cd, ci, ci, P]),
        "orc_bicg": (Result, [ci, I, I, D, D, D, ci, cd, ci, ci, P]),
        "orc_bicgstab": (Result, [ci, I, I, D, D, D, ci, cd, ci, ci, P]),
        "orc_gmres": (Result, [ci, I, I, D, D, D, ci, cd, ci, ci, ci, P]),
    }
    for k, (r, a) in sig.items():
        f = getattr(L, k)
        f.restype, f.argtypes = r, a
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- generators
def poisson1d(gn, is_=0, ie=None):
    ie = gn if ie is None else ie
    n = ie - is_
    ptr = np.zeros(n + 1, np.int32)
    idx = np.zeros(3 * n, np.int32)
    val = np.zeros(3 * n, np.float64)
    nnz = lib().orc_gen_poisson1d(gn, is_, ie, ptr, idx, val)
    return ptr, idx[:nnz].copy(), val[:nnz].copy()


def poisson3d(l, m, n, sort_cols=False, is_=0, ie=None):
    gn = l * m * n
    ie = gn if ie is None else ie
    rows = ie - is_
    ptr = np.zeros(rows + 1, np.int32)
    idx = np.zeros(7 * rows, np.int32)
    val = np.zeros(7 * rows, np.float64)
    nnz = lib().orc_gen_poisson3d(l, m, n, is_, ie, int(sort_cols), ptr, idx, val)
    return ptr, idx[:nnz].copy(), val[:nnz].copy()


def random_csr(n, avg_nnz, seed, ncols=None, sort_cols=False, empty_rows=True, long_row=None):
    """Irregular CSR with distinct columns per row, values in [-1,1).  Test input, no reference analogue."""
    rng = np.random.default_rng(seed)
    ncols = n if ncols is None else ncols
    lens = rng.poisson(avg_nnz, n).astype(np.int64)
    lens = np.minimum(lens, ncols)
    if empty_rows and n > 4:
        lens[rng.integers(0, n, max(1, n // 16))] = 0
    if long_row is not None and n > 0:
        lens[rng.integers(0, n)] = min(long_row, ncols)
    ptr = np.zeros(n + 1, np.int32)
    ptr[1:] = np.cumsum(lens)
    idx = np.empty(int(ptr[-1]), np.int32)
    for r in range(n):
        k = int(lens[r])
        if k:
            cols = rng.choice(ncols, k, replace=False).astype(np.int32)
            if sort_cols:
                cols.sort()
            idx[ptr[r]:ptr[r + 1]] = cols
    val = rng.uniform(-1, 1, int(ptr[-1]))
    return ptr, idx, val


# ---------------------------------------------------------------- spmv

def fem3(G, dofs=3):
    """BASELINE config 4 stand-in (SuiteSparse Queen_4147's sparsity class, 3-D structural FEM): `dofs` unknowns per node
    of a G^3 grid, 27-node connectivity -- dofs x dofs blocks, up to 27 * dofs entries per row (boundary rows shorter);
    symmetric, strictly diagonally dominant.  Returns ptr (int32), idx (int32), val, n."""
    nodes = G ** 3
    z, y, x = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    z, y, x = z.ravel(), y.ravel(), x.ravel()
    offs = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    masks = [((z + dz >= 0) & (z + dz < G) & (y + dy >= 0) & (y + dy < G) & (x + dx >= 0) & (x + dx < G)) for dz, dy, dx in offs]
    cnt = np.zeros(nodes, np.int64)
    pos = []
    for m in masks:
        pos.append(cnt.copy())
        cnt += m
    rowlen = np.repeat(dofs * cnt, dofs)
    ptr = np.zeros(dofs * nodes + 1, np.int64)
    np.cumsum(rowlen, out=ptr[1:])
    nnz = int(ptr[-1])
    idx = np.empty(nnz, np.int32)
    val = np.empty(nnz)
    for k, ((dz, dy, dx), m) in enumerate(zip(offs, masks)):
        p = np.nonzero(m)[0]
        q = p + (dz * G + dy) * G + dx
        dist = abs(dz) + abs(dy) + abs(dx)
        for d in range(dofs):
            base = ptr[dofs * p + d] + dofs * pos[k][p]
            for e in range(dofs):
                idx[base + e] = dofs * q + e
                if dist == 0:
                    val[base + e] = 0.0 if d == e else -0.125          # diagonal filled below
                else:
                    val[base + e] = -(1.0 if d == e else 0.25) / dist
    n = dofs * nodes
    rowsum = np.add.reduceat(np.abs(val), ptr[:-1])                    # diagonal = 1 + sum of |off-diagonal| of the row
    self_k = offs.index((0, 0, 0))
    p = np.arange(nodes)
    for d in range(dofs):
        at = ptr[dofs * p + d] + dofs * pos[self_k][p] + d
        val[at] = rowsum[dofs * p + d] + 1.0
    return ptr.astype(np.int32), idx, val, n


def heavy_tail(n, seed=3, cap=9000):
    """Heavy-tailed row lengths (Pareto, 1 .. cap: some rows longer than the kernels' LDS stage), random columns (repeats
    allowed), values in [-1,1) with the diagonal entry -- stored first -- raised to 1 + the row's absolute sum: non-symmetric,
    strictly diagonally dominant.  The load-balance stress of BASELINE config 4's class.  Returns ptr, idx, val."""
    rng = np.random.default_rng(seed)
    lens = np.minimum((rng.pareto(1.3, n) * 8 + 2).astype(np.int64), cap)
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[-1])
    idx = rng.integers(0, n, nnz, dtype=np.int32)
    val = rng.uniform(-1, 1, nnz)
    idx[ptr[:-1]] = np.arange(n, dtype=np.int32)                       # first entry of every row: the diagonal
    val[ptr[:-1]] = 0.0
    val[ptr[:-1]] = np.add.reduceat(np.abs(val), ptr[:-1]) + 1.0
    return ptr.astype(np.int32), idx, val

def unstructured_mesh(nodes, seed=7, kmin=6, kmax=22, cells=16):
    """An unstructured 3-D mesh, one unknown per node, varying coefficients, no block structure (round 6: the irregular class none of the plan's special forms catch --
    no repeating row patterns, no dofs-per-node blocks, no constant values).  Points at random in the unit cube; every node is joined to its kmin .. kmax
    nearest neighbours (its own draw) and the graph is symmetrised, which gives ragged rows of 8 .. ~40 entries (mean ~19).  Numbered as a mesher would leave it: along a coarse
    Morton curve (16^3 cells), in random order inside a cell.  Edge weights -1 / distance scaled by a random factor in [0.5, 1.5) -- the same on both sides: symmetric;
    diagonal = 1.02 x the row's absolute sum: positive definite.  Columns ascending within a row.  Deterministic (numpy + scipy.spatial).  Returns ptr, idx, val."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((nodes, 3))
    cell = np.minimum((pts * cells).astype(np.int64), cells - 1)      # (cells = 16: the 4096 Morton cells of the goldens; fewer cells = a numbering with less locality, perf probes)
    morton = np.zeros(nodes, np.int64)
    for bit in range(4):
        for d in range(3):
            morton |= ((cell[:, d] >> bit) & 1) << (3 * bit + d)
    order = np.lexsort((rng.random(nodes), morton))
    pts = pts[order]
    _, nb = cKDTree(pts).query(pts, k=kmax + 1, workers=min(16, os.cpu_count() or 1))      # (the same neighbours whatever the number of workers)
    want = rng.integers(kmin, kmax + 1, nodes)                         # every node asks for its own number of neighbours: ragged rows
    keep = np.arange(1, kmax + 1)[None, :] <= want[:, None]
    src = np.repeat(np.arange(nodes, dtype=np.int64), kmax)[keep.reshape(-1)]
    dst = nb[:, 1:].reshape(-1).astype(np.int64)[keep.reshape(-1)]
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    edges = np.unique(lo * nodes + hi)                                  # undirected edges, each once, in a fixed order
    lo, hi = edges // nodes, edges % nodes
    w = -(0.5 + rng.random(len(edges))) / np.linalg.norm(pts[lo] - pts[hi], axis=1)
    rows = np.concatenate([lo, hi, np.arange(nodes)])
    cols = np.concatenate([hi, lo, np.arange(nodes)])
    absum = np.bincount(np.concatenate([lo, hi]), weights=np.abs(np.concatenate([w, w])), minlength=nodes)
    vals = np.concatenate([w, w, 1.02 * absum])                         # 2 % of dominance: GMRES(30) / CG / BiCGSTAB need ~130 / 120 / 70 iterations at 60 000 nodes
    at = np.lexsort((cols, rows))
    rows, cols, vals = rows[at], cols[at], vals[at]
    ptr = np.zeros(nodes + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=nodes), out=ptr[1:])
    return ptr.astype(np.int32), cols.astype(np.int32), np.ascontiguousarray(vals)


def spmv_csr(ptr, idx, val, x):
    n = len(ptr) - 1
    y = np.empty(n)
    lib().orc_spmv_csr(n, ptr, idx, val, x, y)
    return y


def spmv_csc(n, np_, ptr, idx, val, x):
    y = np.empty(n)
    lib().orc_spmv_csc(n, np_, ptr, idx, val, x, y)
    return y


def spmv_ell(n, maxnzr, idx, val, x):
    y = np.empty(n)
    lib().orc_spmv_ell(n, maxnzr, idx, val, x, y)
    return y


def spmv_dia(n, nnd, off, val, x, nchunks=1):
    y = np.empty(n)
    lib().orc_spmv_dia(n, nnd, nchunks, off, val, x, y)
    return y


def spmv_jad(n, maxnzr, perm, ptr, idx, val, x, nchunks=1):
    y = np.empty(n)
    lib().orc_spmv_jad(n, maxnzr, nchunks, perm, ptr, idx, val, x, y)
    return y


def spmv_bsr(n, nr, bnr, bnc, bptr, bidx, val, x):
    y = np.zeros(nr * bnr)
    xx = np.zeros(max(len(x), (int(bidx.max()) + 1) * bnc if len(bidx) else 0))
    xx[:len(x)] = x
    lib().orc_spmv_bsr(n, nr, bnr, bnc, bptr, bidx, val, xx, y)
    return y[:n].copy()


def spmvh_csr(ptr, idx, val, x, np_=None):
    n = len(ptr) - 1
    np_ = n if np_ is None else np_
    y = np.empty(np_)
    lib().orc_spmvh_csr(n, np_, ptr, idx, val, x, y)
    return y


def spmvh_csc(np_, ptr, idx, val, x):
    y = np.empty(np_)
    lib().orc_spmvh_csc(np_, ptr, idx, val, x, y)
    return y


def spmvh_ell(n, maxnzr, idx, val, x):
    y = np.empty(n)
    lib().orc_spmvh_ell(n, n, maxnzr, idx, val, x, y)
    return y


def spmvh_dia(n, nnd, off, val, x):
    y = np.empty(n)
    lib().orc_spmvh_dia(n, n, nnd, off, val, x, y)
    return y


def spmvh_jad(n, maxnzr, perm, ptr, idx, val, x):
    y = np.empty(n)
    lib().orc_spmvh_jad(n, n, maxnzr, perm, ptr, idx, val, x, y)
    return y


def spmvh_bsr(n, nr, bnr, bnc, bptr, bidx, val, x):
    ylen = max(nr * bnr, (int(bidx.max()) + 1) * bnc if len(bidx) else 0)
    y = np.zeros(ylen)
    xx = np.zeros(nr * bnr)
    xx[:n] = x[:n]
    lib().orc_spmvh_bsr(nr, bnr, bnc, bptr, bidx, val, xx, y, ylen)
    return y[:n].copy()


# ---------------------------------------------------------------- conversions
def csr2ell(ptr, idx, val):
    n = len(ptr) - 1
    mx = lib().orc_ell_maxnzr(n, ptr)
    eidx = np.empty(mx * n, np.int32)
    eval_ = np.empty(mx * n)
    lib().orc_csr2ell(n, ptr, idx, val, mx, eidx, eval_)
    return mx, eidx, eval_


def csr2csc(ptr, idx, val, np_=None):
    n = len(ptr) - 1
    np_ = n if np_ is None else np_
    cptr = np.empty(np_ + 1, np.int32)
    cidx = np.empty(len(idx), np.int32)
    cval = np.empty(len(idx))
    lib().orc_csr2csc(n, np_, ptr, idx, val, cptr, cidx, cval)
    return cptr, cidx, cval


def sort_rows(ptr, idx, val):
    idx = idx.copy()
    val = val.copy()
    for r in range(len(ptr) - 1):
        s, e = ptr[r], ptr[r + 1]
        o = np.argsort(idx[s:e], kind="stable")
        idx[s:e] = idx[s:e][o]
        val[s:e] = val[s:e][o]
    return idx, val


def csr2dia(ptr, idx, val):
    """Input rows must be column-sorted (the reference sorts its input in place first)."""
    n = len(ptr) - 1
    nnz = len(idx)
    nnd = lib().orc_csr2dia(n, nnz, ptr, idx, val, None, None)
    off = np.empty(nnd, np.int32)
    dval = np.empty(nnd * n)
    lib().orc_csr2dia(n, nnz, ptr, idx, val, _p(off), _p(dval))
    return nnd, off, dval


def csr2jad(ptr, idx, val):
    n = len(ptr) - 1
    mx = lib().orc_ell_maxnzr(n, ptr)
    perm = np.empty(n, np.int32)
    jptr = np.empty(mx + 1, np.int32)
    jidx = np.empty(len(idx), np.int32)
    jval = np.empty(len(idx))
    lib().orc_csr2jad(n, ptr, idx, val, mx, perm, jptr, jidx, jval)
    return mx, perm, jptr, jidx, jval


def csr2bsr(ptr, idx, val, bnr=2, bnc=2):
    n = len(ptr) - 1
    nr = 1 + (n - 1) // bnr
    bptr = np.empty(nr + 1, np.int32)
    bnnz = lib().orc_csr2bsr(n, ptr, idx, val, bnr, bnc, bptr, None, None)
    bidx = np.empty(bnnz, np.int32)
    bval = np.empty(bnnz * bnr * bnc)
    lib().orc_csr2bsr(n, ptr, idx, val, bnr, bnc, bptr, _p(bidx), _p(bval))
    return nr, bptr, bidx, bval


def csr_diagonal(ptr, idx, val):
    n = len(ptr) - 1
    d = np.empty(n)
    lib().orc_csr_diagonal(n, ptr, idx, val, d)
    return d


# ---------------------------------------------------------------- solvers
def _solve(fn, ptr, idx, val, b, x0, precon, tol, maxiter, extra):
    n = len(ptr) - 1
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    rh = np.zeros(maxiter + 2)
    pre = {"none": 0, "jacobi": 1}[precon]
    args = [n, ptr, idx, val, np.ascontiguousarray(b), x, pre, float(tol), int(maxiter)]
    args += extra + [int(x0 is None), _p(rh)]
    res = fn(*args)
    return x, res.iter, res.retcode, res.resid, rh


def cg(ptr, idx, val, b, x0=None, precon="none", tol=1e-12, maxiter=1000):
    return _solve(lib().orc_cg, ptr, idx, val, b, x0, precon, tol, maxiter, [])


def bicg(ptr, idx, val, b, x0=None, precon="none", tol=1e-12, maxiter=1000):
    return _solve(lib().orc_bicg, ptr, idx, val, b, x0, precon, tol, maxiter, [])


def bicgstab(ptr, idx, val, b, x0=None, precon="none", tol=1e-12, maxiter=1000):
    return _solve(lib().orc_bicgstab, ptr, idx, val, b, x0, precon, tol, maxiter, [])


def gmres(ptr, idx, val, b, x0=None, precon="none", tol=1e-12, maxiter=1000, restart=40):
    return _solve(lib().orc_gmres, ptr, idx, val, b, x0, precon, tol, maxiter, [int(restart)])


# ---------------------------------------------------------------- split form A = L + D + U
# Test infrastructure in plain Python / numpy (small cases): which terms the reference's is_splited branches add to a row, in
# which order, restated from src/matvec/lis_matvec_<fmt>.c and src/matrix/lis_matrix_<fmt>.c (split routines).  Pinned against
# the reference itself by tests/golden/make_golden_split.py -> tests/golden/split_golden.npz.
def _chain(first_from_zero, terms):
    """t = first product (or 0.0 + first product), then t += each further product, one rounding per operation"""
    t = 0.0 if first_from_zero else None
    for v, xv in terms:
        p = float(v) * float(xv)
        t = p if t is None else t + p
    return t


def split_csr(ptr, idx, val):
    """lis_matrix_split_csr (lis_matrix_csr.c:765): per row the entries left / right of the diagonal, D = the last entry on it"""
    n = len(ptr) - 1
    L, U, D = [[] for _ in range(n)], [[] for _ in range(n)], np.zeros(n)
    for r in range(n):
        for k in range(ptr[r], ptr[r + 1]):
            (L[r] if idx[k] < r else U[r] if idx[k] > r else []).append((idx[k], val[k]))
            if idx[k] == r:
                D[r] = val[k]
    return L, U, D


def spmv_split_csr(ptr, idx, val, x):
    """lis_matvec_csr.c:64-89"""
    L, U, D = split_csr(ptr, idx, val)
    return np.array([_chain(False, [(D[r], x[r])] + [(v, x[c]) for c, v in L[r]] + [(v, x[c]) for c, v in U[r]]) for r in range(len(D))])


def spmv_split_csc(n, cptr, cidx, cval, x):
    """lis_matrix_split_csc (lis_matrix_csc.c:493) + lis_matvec_csc.c:65-90: y = D x, then column by column its "L" entries
    (row index < column) and its "U" entries (row index > column), scattered into y"""
    D = np.zeros(n)
    terms = [[] for _ in range(n)]
    for c in range(len(cptr) - 1):
        lo = [(cidx[k], cval[k]) for k in range(cptr[c], cptr[c + 1]) if cidx[k] < c]
        up = [(cidx[k], cval[k]) for k in range(cptr[c], cptr[c + 1]) if cidx[k] > c]
        for k in range(cptr[c], cptr[c + 1]):
            if cidx[k] == c:
                D[c] = cval[k]
        for r, v in lo + up:
            terms[r].append((v, x[c]))
    return np.array([_chain(False, [(D[r], x[r])] + terms[r]) for r in range(n)])


def spmv_split_ell(n, maxnzr, eidx, eval_, x):
    """lis_matrix_split_ell (lis_matrix_ell.c:324) + lis_matvec_ell.c:56-87: L / U are ELL arrays padded to their own widths
    with (value 0, column = row); D takes a diagonal entry only when its value is not 0"""
    rows_l, rows_u, D = [], [], np.zeros(n)
    for r in range(n):
        ent = [(eidx[j * n + r], eval_[j * n + r]) for j in range(maxnzr)]
        rows_l.append([(c, v) for c, v in ent if c < r])
        rows_u.append([(c, v) for c, v in ent if c > r])
        for c, v in ent:
            if c == r and v != 0.0:
                D[r] = v
    lmax, umax = max((len(t) for t in rows_l), default=0), max((len(t) for t in rows_u), default=0)
    y = []
    for r in range(n):
        lo = rows_l[r] + [(r, 0.0)] * (lmax - len(rows_l[r]))
        up = rows_u[r] + [(r, 0.0)] * (umax - len(rows_u[r]))
        y.append(_chain(False, [(D[r], x[r])] + [(v, x[c]) for c, v in lo] + [(v, x[c]) for c, v in up]))
    return np.array(y)


def spmv_split_dia(n, nnd, off, dval, x):
    """lis_matrix_split_dia (lis_matrix_dia.c:782) + lis_matvec_dia.c:56-123: whole diagonals move to L (offset < 0), U (> 0), D"""
    y = []
    for r in range(n):
        d0 = [dval[d * n + r] for d in range(nnd) if off[d] == 0]
        terms = [(d0[-1] if d0 else 0.0, x[r])]
        for sign in (-1, 1):
            terms += [(dval[d * n + r], x[r + off[d]]) for d in range(nnd) if off[d] * sign > 0 and 0 <= r + off[d] < n]
        y.append(_chain(False, terms))
    return np.array(y)


def spmv_split_bsr(n, nr, bnr, bnc, bptr, bidx, bval, x):
    """lis_matrix_split_bsr (lis_matrix_bsr.c:1131, square blocks) + lis_matvec_bsr_NxN (lis_matvec_bsr.c:159 / :293 / :453 / :644):
    the block row of D column by column -- the first product starts the sum --, then the L blocks, then the U blocks; blocks
    larger than 4 x 4 take the generic routine (:70-118), which starts every sum at 0.0"""
    assert bnr == bnc
    bs = bnr * bnc
    xp = np.concatenate([x, np.zeros(nr * bnr + bnc - len(x))])
    y = np.zeros(nr * bnr)
    for bi in range(nr):
        blocks = [(bidx[k], bval[k * bs:(k + 1) * bs]) for k in range(bptr[bi], bptr[bi + 1])]
        dblk = np.zeros(bs)
        for c, v in blocks:
            if c == bi:
                dblk = v
        order = [(bi, dblk)] + [(c, v) for c, v in blocks if c < bi] + [(c, v) for c, v in blocks if c > bi]
        for ii in range(bnr):
            terms = [(v[j * bnr + ii], xp[c * bnc + j]) for c, v in order for j in range(bnc)]
            y[bi * bnr + ii] = _chain(bnr > 4, terms)
    return y[:n]


def spmv_split_jad(n, maxnzr, perm, jptr, jidx, jval, x):
    """lis_matvec_jad.c:60-140: y = D x; w = sum over the L entries (from 0, jagged-diagonal order); y[row] += w; the same for U.
    The jagged order of a row's L (U) entries is their order in A's jagged diagonals (lis_matrix_split_jad keeps it)."""
    lo, up, D = [[] for _ in range(n)], [[] for _ in range(n)], np.zeros(n)
    for j in range(maxnzr):
        for s, k in enumerate(range(jptr[j], jptr[j + 1])):
            r = perm[s]
            if jidx[k] < r:
                lo[r].append((jval[k], x[jidx[k]]))
            elif jidx[k] > r:
                up[r].append((jval[k], x[jidx[k]]))
            else:
                D[r] = jval[k]
    return np.array([(float(D[r]) * float(x[r]) + _chain(True, lo[r])) + _chain(True, up[r]) for r in range(n)])
