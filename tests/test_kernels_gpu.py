"""Parity of the hand-written HIP kernels (kernel-level C ABI, include/liship.h) against the CPU oracle.

Bit-exact for every SpMV format and every element-wise kernel (same rounding sequence as the
reference's loops); reductions are compared at 1e-14 relative (tree order differs from the reference's
left-to-right order -- tolerance stated here, SURVEY 7 "hard parts").
"""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import lis_amd
from lis_amd import DeviceArray as DA, check

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lis_ref_golden.npz"))


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    return lib


def dev_csr_spmv(lib, ptr, idx, val, x, variant=0, rows=None):
    n = len(ptr) - 1
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx = DA.from_host(x, np.float64)
    dy = DA.from_host(np.full(n, np.nan), np.float64)
    plan = C.c_void_p()
    lib.liship_spmv_csr_set_variant(variant)          # geometry bits are read at plan creation
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    if rows is None:
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
    else:
        check(lib.liship_spmv_csr_rows_f64(plan, rows[0], rows[1], dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
    lib.liship_spmv_csr_set_variant(0)
    y = dy.to_host()
    check(lib.liship_csr_plan_destroy(plan))
    return y


CSR_CASES = {
    "p1d_10000": lambda: orc.poisson1d(10000),
    "p3d_16": lambda: orc.poisson3d(16, 16, 16),
    "p3d_20x17x13_sorted": lambda: orc.poisson3d(20, 17, 13, sort_cols=True),
    "p3d_64": lambda: orc.poisson3d(64, 64, 64),
    "rand_5000": lambda: orc.random_csr(5000, 11, seed=1),
    "rand_long_rows": lambda: orc.random_csr(300, 40, seed=2, ncols=9000, long_row=7001),
    "rand_wide_77": lambda: orc.random_csr(4000, 77, seed=3, ncols=4000, empty_rows=False),
    "mostly_empty": lambda: orc.random_csr(20000, 0.05, seed=4),
    "single_row": lambda: orc.random_csr(1, 5, seed=5, ncols=64, empty_rows=False),
}


@pytest.fixture(autouse=True)
def _chain_mode_for_the_long_row_cases(request, lib):
    """Round 6: the part of a row beyond the LDS stage is added by a workgroup tree BY DEFAULT when it is at least 1024 entries long (value to 1e-14 of the row's
    magnitude, not the reference's last bits: test_spmv_csr_long_row_tree_is_the_default, test_spmv_csr_long_rows_in_the_default_mode).  The parametrised cases
    whose matrices hold such rows demand the reference's BITS, so they run in the mode that promises them: the left-to-right chain (LIS_AMD_LONG_ROW_CHAIN=1)."""
    cid = getattr(request.node, "callspec", None)
    names = [str(v) for v in cid.params.values()] if cid else []
    chain = any(n in ("rand_long_rows", "fem3_long_row", "long_rows") for n in names)
    if chain:
        check(lib.liship_spmv_csr_set_long_row_tree(0))
    yield
    if chain:
        check(lib.liship_spmv_csr_set_long_row_tree(1))


# liship_spmv_csr_set_variant bits (lis_amd/csrc/kernels/spmv_csr.hip): 0 = shipped row-gather kernel with LDS-DMA;
# 0x2 / 0x4 products kernel (scalar / vector loads); 0x10 / 0x50 the 256 / 2048 and 512 / 4096 geometries; 0x1000000 unaligned row blocks.  Every value selects kernels that give the reference's bits.
VARIANTS = [0x0, 0x2, 0x4, 0x10, 0x12, 0x14, 0x50, 0x54, 0x1000000, 0x1000004]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", list(CSR_CASES))
def test_spmv_csr_bit_exact(lib, name, variant):
    ptr, idx, val = CSR_CASES[name]()
    ncols = max(len(ptr) - 1, int(idx.max()) + 1 if len(idx) else 1)
    x = np.random.default_rng(3).uniform(-1, 1, ncols)
    y = dev_csr_spmv(lib, ptr, idx, val, x, variant)
    assert np.array_equal(y, orc.spmv_csr(ptr, idx, val, x)), name


@pytest.mark.parametrize("variant", VARIANTS)
def test_spmv_csr_long_rows_in_the_default_mode(lib, variant):
    """the matrix of the chain-mode cases (a 7001-entry row among rows of ~40) in the DEFAULT mode, through every kernel variant: rows shorter than 1024 entries carry
    the reference's bits, the long row its value to 1e-14 of the sum of its terms' magnitudes, and two runs agree bit for bit"""
    assert lib.liship_spmv_csr_switches() & 4
    ptr, idx, val = CSR_CASES["rand_long_rows"]()
    x = np.random.default_rng(3).uniform(-1, 1, max(len(ptr) - 1, int(idx.max()) + 1))
    ref = orc.spmv_csr(ptr, idx, val, x)
    y1, y2 = dev_csr_spmv(lib, ptr, idx, val, x, variant), dev_csr_spmv(lib, ptr, idx, val, x, variant)
    assert np.array_equal(y1.view(np.uint64), y2.view(np.uint64))
    lens = np.diff(ptr)
    short = lens < 1024
    assert (~short).sum() == 1 and np.array_equal(y1[short], ref[short])
    absum = np.add.reduceat(np.abs(val * x[idx]), np.minimum(ptr[:-1], len(val) - 1).astype(np.int64)) * (lens > 0)
    assert np.all(np.abs(y1 - ref) <= 1e-14 * absum)


def test_spmv_csr_empty_matrix_and_zero_rows(lib):
    ptr = np.zeros(1001, np.int32)
    y = dev_csr_spmv(lib, ptr, np.zeros(0, np.int32), np.zeros(0), np.ones(1000))
    assert np.array_equal(y, np.zeros(1000))


def test_spmv_csr_special_values(lib):
    # signed zeros, infinities and NaN propagate exactly as in the reference loop
    ptr, idx, val = orc.random_csr(500, 6, seed=8)
    x = np.random.default_rng(1).uniform(-1, 1, 500)
    x[::7] = 0.0
    x[3::11] = -0.0
    val[::13] = -0.0
    x[50] = np.inf
    x[60] = np.nan
    y = dev_csr_spmv(lib, ptr, idx, val, x)
    ref = orc.spmv_csr(ptr, idx, val, x)
    assert np.array_equal(y.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("kind", ["box27_constant", "box27_varying", "p3d_constant", "ell_padded_rows"])
def test_special_values_through_the_planned_kernels(lib, kind):
    """signed zeros, infinities and NaN in x (and -0.0 / 0.0 among the values) through the kernels the plan chooses -- the staged wide-record kernel, four lanes
    per row with x staged, the dominant-pattern value-record kernel, and the dominant pattern's padded rows (an ELL row form: trailing (row, +0.0) entries,
    where 0.0 * inf must stay NaN) -- uint64-equal to the reference loop, masked -0.0 terms and speculative loads included"""
    if kind.startswith("box27"):
        ptr, idx, val = stencil_box((14, 12, 20))
        if kind == "box27_varying":
            val = val * np.random.default_rng(4).uniform(0.5, 1.5, len(val))
            val[::17] = -0.0
    elif kind == "p3d_constant":
        ptr, idx, val = orc.poisson3d(8, 8, 256, sort_cols=True)
    else:                                                   # every row padded to 7 entries with (row, +0.0) behind its own, as lis_matrix_convert_csr2ell lays them out
        p0, i0, v0 = orc.poisson3d(8, 8, 256, sort_cols=True)
        n0 = len(p0) - 1
        ptr = np.arange(0, 7 * n0 + 1, 7, dtype=np.int32)
        idx, val = np.empty(7 * n0, np.int32), np.zeros(7 * n0)
        for r in range(n0):
            k = p0[r + 1] - p0[r]
            idx[7 * r:7 * r + k], val[7 * r:7 * r + k] = i0[p0[r]:p0[r + 1]], v0[p0[r]:p0[r + 1]]
            idx[7 * r + k:7 * r + 7] = r
    n = len(ptr) - 1
    x = np.random.default_rng(2).uniform(-1, 1, n)
    x[::7] = 0.0
    x[3::11] = -0.0
    for pos, v in ((0, np.inf), (n - 1, -np.inf), (n // 2, np.nan), (n // 3, np.inf), (257, np.nan), (5, -np.inf)):
        x[pos] = v
    ref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    if kind == "box27_constant":
        assert lib.liship_csr_plan_wide_dominant(plan) == 1
    if kind == "box27_varying":
        assert lib.liship_csr_plan_team_form(plan) == 2
    if kind in ("p3d_constant", "ell_padded_rows"):
        assert lib.liship_csr_plan_dominant_pattern(plan) == 1
    nanpos = np.isnan(ref)
    for variant in (0, 0x4000, 0x2000, 0x20000000):
        lib.liship_spmv_csr_set_variant(variant)
        dy = DA.from_host(np.full(n, 7.0), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        y = dy.to_host()
        assert np.array_equal(np.isnan(y), nanpos), hex(variant)                      # (a NaN's payload is the hardware's: positions, and every other bit)
        assert np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), hex(variant)
    lib.liship_spmv_csr_set_variant(0)
    check(lib.liship_csr_plan_destroy(plan))


def test_spmv_csr_row_range(lib):
    ptr, idx, val = orc.poisson3d(12, 12, 12)
    n = len(ptr) - 1
    x = np.random.default_rng(3).uniform(-1, 1, n)
    ref = orc.spmv_csr(ptr, idx, val, x)
    for rb, re in [(0, 144), (144, n - 144), (n - 144, n), (7, 9), (100, 100)]:
        y = dev_csr_spmv(lib, ptr, idx, val, x, rows=(rb, re))
        assert np.array_equal(y[rb:re], ref[rb:re])
        assert np.all(np.isnan(y[:rb])) and np.all(np.isnan(y[re:]))


@pytest.mark.parametrize("name", ["p1d100", "p3d_6x5x4", "p3d_8s", "irr150"])
def test_golden_all_formats(lib, name):
    """Every format against the vectors the reference itself produced (tests/golden)."""
    ptr, idx, val, x = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val", "x"))
    n = len(ptr) - 1
    assert np.array_equal(dev_csr_spmv(lib, ptr, idx, val, x), G[f"{name}/y_csr"])
    dx = DA.from_host(x)
    dy = DA.zeros(n + 8, np.float64)

    def run(fn, *args):
        check(lib.liship_memset(dy.ptr, 0xFF, dy.nbytes, None))
        check(fn(*args, dx.ptr, dy.ptr, None))
        return dy.to_host(n)

    g = lambda k: G[f"{name}/{k}"]
    eidx, ev = DA.from_host(g("ell/index")), DA.from_host(g("ell/value"))
    assert np.array_equal(run(lib.liship_spmv_ell_f64, n, int(g("ell/maxnzr")[0]), eidx.ptr, ev.ptr), g("y_ell"))
    off, dv = DA.from_host(g("dia/index")), DA.from_host(g("dia/value"))
    assert np.array_equal(run(lib.liship_spmv_dia_f64, n, n, int(g("dia/nnd")[0]), off.ptr, dv.ptr), g("y_dia"))
    perm, jp, ji, jv = (DA.from_host(g(f"jad/{k}")) for k in ("row", "ptr", "index", "value"))
    assert np.array_equal(run(lib.liship_spmv_jad_f64, n, int(g("jad/maxnzr")[0]), perm.ptr, jp.ptr, ji.ptr, jv.ptr), g("y_jad"))
    bp, bi, bv = (DA.from_host(g(f"bsr/{k}")) for k in ("bptr", "bindex", "value"))
    xpad = np.zeros(n + 8)
    xpad[:n] = x
    dx = DA.from_host(xpad)
    assert np.array_equal(run(lib.liship_spmv_bsr_f64, int(g("bsr/nr")[0]), 2, 2, bp.ptr, bi.ptr, bv.ptr), g("y_bsr"))
    # CSC: the device keeps the column-ordered transpose as CSR; its row sums are the reference's CSC sums
    cptr, cidx, cval = g("csc/ptr"), g("csc/index"), g("csc/value")
    order = np.lexsort((np.repeat(np.arange(n), np.diff(cptr)), cidx))    # stable by row, column ascending
    tptr = np.zeros(n + 1, np.int32)
    np.add.at(tptr, cidx + 1, 1)
    tptr = np.cumsum(tptr).astype(np.int32)
    tidx = np.repeat(np.arange(n, dtype=np.int32), np.diff(cptr))[order]
    assert np.array_equal(dev_csr_spmv(lib, tptr, tidx, cval[order], x), g("y_csc"))


@pytest.mark.parametrize("fmt", ["ell", "dia", "jad", "bsr"])
@pytest.mark.parametrize("case", ["p3d_20x17x13_sorted", "rand_5000"])
def test_formats_vs_oracle(lib, fmt, case):
    ptr, idx, val = CSR_CASES[case]()
    n = len(ptr) - 1
    x = np.random.default_rng(9).uniform(-1, 1, n + 8)
    x[n:] = 0.0
    dx, dy = DA.from_host(x), DA.zeros(n + 8, np.float64)
    if fmt == "ell":
        mx, eidx, ev = orc.csr2ell(ptr, idx, val)
        a, b = DA.from_host(eidx), DA.from_host(ev)
        check(lib.liship_spmv_ell_f64(n, mx, a.ptr, b.ptr, dx.ptr, dy.ptr, None))
        ref = orc.spmv_ell(n, mx, eidx, ev, x[:n].copy())
    elif fmt == "dia":
        if case.startswith("rand"):
            pytest.skip("DIA of an irregular matrix is dense")
        sidx, sval = orc.sort_rows(ptr, idx, val)
        nnd, off, dv = orc.csr2dia(ptr, sidx, sval)
        a, b = DA.from_host(off), DA.from_host(dv)
        check(lib.liship_spmv_dia_f64(n, n, nnd, a.ptr, b.ptr, dx.ptr, dy.ptr, None))
        ref = orc.spmv_dia(n, nnd, off, dv, x[:n].copy())
    elif fmt == "jad":
        mx, perm, jptr, jidx, jv = orc.csr2jad(ptr, idx, val)
        a, b, c, d = DA.from_host(perm), DA.from_host(jptr), DA.from_host(jidx), DA.from_host(jv)
        check(lib.liship_spmv_jad_f64(n, mx, a.ptr, b.ptr, c.ptr, d.ptr, dx.ptr, dy.ptr, None))
        ref = orc.spmv_jad(n, mx, perm, jptr, jidx, jv, x[:n].copy())
    else:
        nr, bptr, bidx, bv = orc.csr2bsr(ptr, idx, val, 3, 2)
        a, b, c = DA.from_host(bptr), DA.from_host(bidx), DA.from_host(bv)
        check(lib.liship_spmv_bsr_f64(nr, 3, 2, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, None))
        ref = orc.spmv_bsr(n, nr, 3, 2, bptr, bidx, bv, x[:n].copy())
    assert np.array_equal(dy.to_host(n), ref)


def banded(n, offsets, seed, ncols=None):
    """rows with one entry on each of the given diagonals that falls inside the matrix, in the given (unsorted) order"""
    rng = np.random.default_rng(seed)
    ncols = ncols or n
    rows, cols = [], []
    for r in range(n):
        for o in offsets:
            if 0 <= r + o < ncols:
                rows.append(r); cols.append(r + o)
    ptr = np.zeros(n + 1, np.int32)
    np.add.at(ptr, np.asarray(rows) + 1, 1)
    return np.cumsum(ptr).astype(np.int32), np.asarray(cols, np.int32), rng.uniform(-1, 1, len(cols))


def cyclic_diagonals(n, count, seed):
    """two entries per row: the main diagonal and one of `count - 1` other diagonals, chosen cyclically"""
    rng = np.random.default_rng(seed)
    others = [5 * k + 1 if k % 2 else -(5 * k + 2) for k in range(count - 1)]
    ptr, cols = [0], []
    for r in range(n):
        o = others[r % len(others)]
        cols += [r] + ([r + o] if 0 <= r + o < n else [])
        ptr.append(len(cols))
    return np.asarray(ptr, np.int32), np.asarray(cols, np.int32), rng.uniform(-1, 1, len(cols))


# row patterns the plan must find on top of the codes: 3 for the 1-D stencil (first row, interior, last row), 27 for the 3-D one
ROW_PATTERNS = {"p1d_10000": 3, "p3d_20x17x13": 27, "p3d_64_sorted": 27, "p3d_40_sorted": 27, "p3d_odd_33x7x5": 27, "diagonals_255": 255,
                "diagonals_300": 0, "rand_5000": 0, "wide_77": 0,        # diagonals_255: 254 two-entry rows + the rows whose second entry falls outside
                "star13_varcoef_20x18x30": 125, "p1d_9999": 3, "p3d_holes": 27, "p3d_varcoef": 27, "p3d_dirichlet": 27, "box27_18x15x13": 27, "box9_70x50": 9}            # p3d_holes: the empty row is one more pattern, a corner row that lost its entries one fewer
# ... and whether it also keeps them as 32 B records (1..7 offsets per pattern, at most 64 patterns: spmv_csr_pattern7_kernel)
# ... and whether every row of a pattern also carries the same values (then the records hold them: spmv_csr_valuerec_kernel)
VALUE_RECORDS = {"p1d_10000": 1, "p1d_9999": 1, "p3d_20x17x13": 1, "p3d_64_sorted": 1, "p3d_40_sorted": 1, "p3d_odd_33x7x5": 1,
                 "p3d_varcoef": 0, "p3d_holes": 0, "diagonals_255": 0, "band_9_unsorted": 0,
                 "p3d_dirichlet": 1,         # ... after the offset patterns were split by the values their rows carry
                 "box27_18x15x13": 2, "box9_70x50": 2, "box27_dirichlet": 2, "box27_varcoef_21x10x9": 0, "box27_varcoef_7x6x70": 0,
                 "box9_varcoef_130x77": 0}
PATTERN_RECORDS = {"p1d_10000": 1, "p1d_9999": 1, "p3d_20x17x13": 1, "p3d_64_sorted": 1, "p3d_40_sorted": 1, "p3d_odd_33x7x5": 1,
                   "diagonals_255": 0, "p3d_holes": 0, "band_9_unsorted": 0, "rand_5000": 0, "p3d_varcoef": 1, "p3d_dirichlet": 1,
                   "box27_18x15x13": 0, "box9_70x50": 0}


def poisson3d_variable_coefficients(nx, ny, nz, seed):
    """the 3-D stencil's pattern with random values: pattern records, but no value records"""
    ptr, idx, val = orc.poisson3d(nx, ny, nz)
    return ptr, idx, np.random.default_rng(seed).uniform(-1, 1, len(val))


def stencil_box(dims, centre=None):
    """the full box stencil on a grid (9 points in 2-D, 27 in 3-D), constant coefficients: -1 on the neighbours, their count on the
    diagonal, Dirichlet truncation; sorted columns.  What the reference's spmvtest2b / spmvtest3b drivers generate."""
    dims = tuple(dims)
    n = int(np.prod(dims))
    grids = np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")
    co = [g.ravel() for g in grids]
    strides = [int(np.prod(dims[k + 1:])) for k in range(len(dims))]
    rows, cols, vals = [], [], []
    import itertools
    for d in itertools.product((-1, 0, 1), repeat=len(dims)):
        m = np.ones(n, bool)
        for k, dk in enumerate(d):
            m &= (co[k] + dk >= 0) & (co[k] + dk < dims[k])
        r = np.nonzero(m)[0]
        rows.append(r)
        cols.append(r + sum(dk * st for dk, st in zip(d, strides)))
        vals.append(np.full(len(r), float(3 ** len(dims) - 1) if centre is None and not any(d) else (centre if not any(d) else -1.0)))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((cols, rows))
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ptr[1:])
    return ptr.astype(np.int32), cols[order].astype(np.int32), vals[order]


def stencil_box_variable_coefficients(dims, seed):
    """the box stencil's pattern with random values (a 27-point discretisation on a non-uniform mesh): rows of 8..27 entries, values
    streamed -- the four-lanes-per-row kernel (spmv_csr_pattern_team_kernel)"""
    ptr, idx, val = stencil_box(dims)
    return ptr, idx, np.random.default_rng(seed).uniform(-1, 1, len(val))


def stencil_points(dims, points, seed):
    """any stencil on a 3-D grid: `points` = (dz, dy, dx) offsets, Dirichlet truncation, sorted columns, random values"""
    dims = tuple(dims)
    n = int(np.prod(dims))
    z, y, x = (g.ravel() for g in np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"))
    rows, cols = [], []
    for dz, dy, dx in points:
        m = (z + dz >= 0) & (z + dz < dims[0]) & (y + dy >= 0) & (y + dy < dims[1]) & (x + dx >= 0) & (x + dx < dims[2])
        r = np.nonzero(m)[0]
        rows.append(r)
        cols.append(r + (dz * dims[1] + dy) * dims[2] + dx)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((cols, rows))
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ptr[1:])
    return ptr.astype(np.int32), cols[order].astype(np.int32), np.random.default_rng(seed).uniform(-1, 1, len(cols))


POINTS_19 = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if abs(dz) + abs(dy) + abs(dx) <= 2]     # faces and edges, no corners
POINTS_STAR13 = [(0, 0, 0)] + [tuple(s * k if a == ax else 0 for a in range(3)) for ax in range(3) for k in (1, 2) for s in (-1, 1)]     # the fourth-order star in 3-D
POINTS_STAR9 = [(0, 0, 0)] + [tuple(s * k if a == ax else 0 for a in range(3)) for ax in (1, 2) for k in (1, 2) for s in (-1, 1)]     # the fourth-order star in 2-D


def stencil_box_slab_with_ghost_plane(dims, seed):
    """the first dims[0] - 1 planes of the box stencil on `dims` with random values: the rows of the last plane are gone, the columns that point into
    it remain (a rank's slab of a row-block partition, ghost columns behind the owned ones): n < ncols, and the rows of the last owned plane carry the
    interior pattern with columns >= n"""
    ptr, idx, val = stencil_box_variable_coefficients(dims, seed)
    keep = (dims[0] - 1) * int(np.prod(dims[1:]))
    return ptr[:keep + 1].copy(), idx[:ptr[keep]].copy(), val[:ptr[keep]].copy()


def stencil_box_with_foreign_rows(dims, seed, every):
    """the box stencil with random values where every `every`-th row moves its last entry 5 columns to the right (or left, at the end): patterns whose
    offsets the interior pattern's runs do not hold -- the rows that gather for themselves in the staged-x kernel"""
    ptr, idx, val = stencil_box_variable_coefficients(dims, seed)
    n = len(ptr) - 1
    idx = idx.copy()
    for r in range(3, n, every):
        k = ptr[r + 1] - 1
        idx[k] = idx[k] + 5 if idx[k] + 5 < n else max(0, idx[ptr[r]] - 9)
        order = np.argsort(idx[ptr[r]:ptr[r + 1]], kind="stable")           # (rows stay sorted; a repeated column is fine for CSR)
        idx[ptr[r]:ptr[r + 1]] = idx[ptr[r]:ptr[r + 1]][order]
        val[ptr[r]:ptr[r + 1]] = val[ptr[r]:ptr[r + 1]][order]
    return ptr, idx, val


def box27_with_dirichlet_rows(dims, every):
    """the 27-point stencil with identity rows stored on the stencil's sparsity (1 on the diagonal, explicit zeros beside it)"""
    ptr, idx, val = stencil_box(dims)
    val = val.copy()
    for r in range(0, len(ptr) - 1, every):
        seg = slice(ptr[r], ptr[r + 1])
        val[seg] = np.where(idx[seg] == r, 1.0, 0.0)
    return ptr, idx, val


def poisson3d_with_dirichlet_rows(nx, ny, nz, every):
    """the 3-D stencil where every `every`-th row is an identity row stored with the stencil's sparsity (1 on the diagonal, explicit
    zeros beside it) -- how boundary conditions are often imposed: the same offset patterns, two value sets for some of them"""
    ptr, idx, val = orc.poisson3d(nx, ny, nz)
    val = val.copy()
    n = len(ptr) - 1
    for r in range(0, n, every):
        seg = slice(ptr[r], ptr[r + 1])
        val[seg] = np.where(idx[seg] == r, 1.0, 0.0)
    return ptr, idx, val


def poisson3d_with_empty_rows(nx, ny, nz, every):
    """the 3-D stencil with every `every`-th row emptied: one more pattern (length 0), which the 32 B records do not take"""
    ptr, idx, val = orc.poisson3d(nx, ny, nz)
    n = len(ptr) - 1
    keep = np.ones(len(idx), bool)
    for r in range(0, n, every):
        keep[ptr[r]:ptr[r + 1]] = False
    cnt = np.diff(ptr)
    cnt[::every] = 0
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), idx[keep], val[keep]


CODED_CASES = {
    "p1d_10000": (lambda: orc.poisson1d(10000), 3),
    "p1d_9999": (lambda: orc.poisson1d(9999), 3),                                           # odd number of values: the last one has no 16 B piece
    "p3d_20x17x13": (lambda: orc.poisson3d(20, 17, 13), 7),
    "p3d_holes": (lambda: poisson3d_with_empty_rows(24, 20, 16, 37), 7),
    "p3d_varcoef": (lambda: poisson3d_variable_coefficients(24, 20, 16, 5), 7),
    "p3d_dirichlet": (lambda: poisson3d_with_dirichlet_rows(24, 20, 16, 11), 7),
    "box27_18x15x13": (lambda: stencil_box((18, 15, 13)), 27),                                  # rows of up to 27 entries: the wide value records
    "box9_70x50": (lambda: stencil_box((70, 50)), 9),
    "box27_dirichlet": (lambda: box27_with_dirichlet_rows((16, 13, 11), 7), 27),              # wide records after the split by values
    "box27_varcoef_21x10x9": (lambda: stencil_box_variable_coefficients((21, 10, 9), 8), 27),      # values streamed: four lanes per row, odd sizes
    "box27_varcoef_7x6x70": (lambda: stencil_box_variable_coefficients((7, 6, 70), 9), 27),
    "box9_varcoef_130x77": (lambda: stencil_box_variable_coefficients((130, 77), 10), 9),
    "s19_varcoef_9x10x21": (lambda: stencil_points((9, 10, 21), POINTS_19, 14), 19),                  # runs of 1 and 3 columns: staged x with uneven runs
    "star13_varcoef_20x18x30": (lambda: stencil_points((20, 18, 30), POINTS_STAR13, 16), 13),         # 125 patterns, 1438 offsets: a table only the team kernels take
    "star9_varcoef_60x70": (lambda: stencil_points((1, 60, 70), POINTS_STAR9, 15), 9),                # a run of 5 and four single columns
    "box27_varcoef_ghost_plane": (lambda: stencil_box_slab_with_ghost_plane((9, 12, 22), 11), 27),     # columns >= n in the last owned plane: the staged loads' clamp
    "box27_varcoef_foreign": (lambda: stencil_box_with_foreign_rows((10, 11, 24), 13, 17), None),       # rows off the interior pattern's runs gather for themselves
    "p3d_64_sorted": (lambda: orc.poisson3d(64, 64, 64, sort_cols=True), 7),
    "band_9_unsorted": (lambda: banded(5000, [40, -1, 0, 1, -40, 3, -3, 900, -900], 1), 9),
    "p3d_40_sorted": (lambda: orc.poisson3d(40, 40, 40, sort_cols=True), 7),                # several row blocks
    "p3d_odd_33x7x5": (lambda: orc.poisson3d(33, 7, 5), 7),                                 # odd number of columns
    "p3d_5x6x256": (lambda: orc.poisson3d(5, 6, 256), 7),                                   # grid lines of 256 rows: the tiled lane -> row mapping, with a tail
    "band_neighbours": (lambda: banded(9000, [-3, -2, -1, 0, 1, 2, 700, 701, -4000], 6), 9),
    "diagonals_255": (lambda: cyclic_diagonals(6000, 255, 2), 255),
    "diagonals_300": (lambda: cyclic_diagonals(6000, 300, 3), 0),                           # too many diagonals
    "band_ghost_columns": (lambda: banded(4000, [-2, 0, 5, 3000, 4100], 4, ncols=8200), 5),      # columns beyond n
    "rand_5000": (lambda: orc.random_csr(5000, 11, seed=1), 0),                              # thousands of offsets
    "mostly_empty": (lambda: orc.random_csr(20000, 0.05, seed=4), None),
    "single_row": (lambda: orc.random_csr(1, 5, seed=5, ncols=64, empty_rows=False), None),
    "wide_77": (lambda: orc.random_csr(4000, 77, seed=3, ncols=4000, empty_rows=False), 0),     # products kernel: never coded
}


# the form of the four-lanes-per-row kernel the plan must choose: 2 = x staged per wavefront (one pattern carries at least half of the rows and its
# offsets are runs of one length), 1 = a gather per entry
TEAM_FORM = {"s19_varcoef_9x10x21": 2, "star9_varcoef_60x70": 2, "star13_varcoef_20x18x30": 2, "box27_varcoef_ghost_plane": 2, "box27_varcoef_foreign": 2, "box27_18x15x13": 2, "box9_70x50": 2, "box27_varcoef_21x10x9": 2, "box27_varcoef_7x6x70": 1, "box9_varcoef_130x77": 2}


# liship_spmv_csr_set_variant bits that select the value-record kernels by hand: 3 the general pattern kernel, 4 the round-2 kernels by
# size, 5 their two-rows-per-lane form, 6 the dominant-pattern kernels in plain form (contiguous chunks; fused dots two rows per lane),
# 7 the default (tiles where the pattern has a stride that 128 divides; fused dots four rows per lane)
VARIANT_OF_FORM = {3: 0x2000, 4: 0x20000000, 5: 0x20004000, 6: 0x10000000, 7: 0, 8: 0x4000}     # 8: the dominant-pattern product with the row blocks' partial sums (dot4)


@pytest.mark.parametrize("name", list(CODED_CASES))
def test_spmv_csr_index_codes(lib, name):
    """one-byte column codes: the plan codes exactly the matrices with <= 255 diagonals, and with the codes every form
    of the product (plain, reduction epilogue, row ranges, parts) returns the bits of the 4 B-index kernel"""
    make, want = CODED_CASES[name]
    ptr, idx, val = make()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1 if len(idx) else 1)
    rng = np.random.default_rng(31)
    x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx if len(idx) else np.zeros(1, np.int32), np.int32), \
        DA.from_host(val if len(val) else np.zeros(1), np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    coded = lib.liship_csr_plan_coded(plan)
    if want is not None:
        assert coded == want, (coded, want)
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    npat = lib.liship_csr_plan_row_patterns(plan)
    assert npat == 0 or coded > 0
    if name in ROW_PATTERNS:
        assert npat == ROW_PATTERNS[name], npat
    if name in PATTERN_RECORDS:
        assert lib.liship_csr_plan_pattern_records(plan) == PATTERN_RECORDS[name]
    if name.startswith(("box", "s19", "star9", "star13")):    # longest pattern of 8..32 offsets: the 144 B records of the four-lanes-per-row kernel
        assert lib.liship_csr_plan_team_records(plan) == 1
    elif lib.liship_csr_plan_pattern_records(plan) == 1:
        assert lib.liship_csr_plan_team_records(plan) == 0
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    if name in VALUE_RECORDS:
        assert lib.liship_csr_plan_value_records(plan) == VALUE_RECORDS[name]
    assert lib.liship_csr_plan_value_records(plan) in (0, 2) or lib.liship_csr_plan_pattern_records(plan) == 1     # 2: the wide records
    results = {}
    for on in (8, 7, 6, 5, 4, 3, 2, 1, 0):     # 4: values in the pattern records too (nothing streamed; 5: the plain product two rows per
        lib.liship_spmv_csr_set_index_codes(1 if on else 0)         # lane, the form for x beyond the Infinity Cache; 6, 7: the dominant pattern's
        lib.liship_spmv_csr_set_row_patterns(1 if on >= 2 else 0)   # gathers speculated, two / four rows per lane), 2: one byte per row (patterns;
        lib.liship_spmv_csr_set_row_values(1 if on >= 4 else 0)     # 3: through the general pattern kernel even when the plan has 32 B records),
        lib.liship_spmv_csr_set_variant(VARIANT_OF_FORM.get(on, 0))  # 1: one byte per non-zero (codes), 0: 4 B indices
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref), on
        out = []
        for sq in (0, 1):
            res = DA.from_host(np.full(2, np.nan), np.float64)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, sq, res.ptr, work.ptr, None)
            if rc == 0:
                assert np.array_equal(dy.to_host(), yref), (on, sq)
                out.append(res.to_host()[:1 + sq].copy())
        if ncols == n:                          # (x, A x): the kernels take x_r from the diagonal's gather when the row has one
            res = DA.from_host(np.full(2, np.nan), np.float64)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dx.ptr, 1, res.ptr, work.ptr, None)
            if rc == 0:
                assert np.array_equal(dy.to_host(), yref), on
                out.append(res.to_host().copy())
        lo, hi = n // 5, n - n // 7
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        for a, b in ((lo, hi), (0, lo), (hi, n)):
            check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref), on
        total, used = 0, C.c_int()
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        ok = True
        for a, b in ((lo, hi), (0, lo), (hi, n)):
            rc = lib.liship_spmv_csr_rows_dot_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1,
                                                  work.ptr, total, C.byref(used), None)
            ok = ok and rc == 0
            total += used.value
        if ok:
            res = DA.from_host(np.full(2, np.nan), np.float64)
            check(lib.liship_spmv_csr_dot_finish_f64(total, 1, res.ptr, work.ptr, None))
            assert np.array_equal(dy.to_host(), yref), on
            out.append(res.to_host().copy())
        results[on] = out
    if lib.liship_csr_plan_team_records(plan):       # four lanes per row: x staged per wavefront (the plan's choice when one pattern dominates) and gathered
        assert lib.liship_csr_plan_team_form(plan) == TEAM_FORM.get(name, lib.liship_csr_plan_team_form(plan)), name
        lib.liship_spmv_csr_set_index_codes(1)
        lib.liship_spmv_csr_set_row_patterns(1)
        lib.liship_spmv_csr_set_row_values(0)
        for variant in (0, 0x4000):          # four lanes per row with x staged; 0x4000: four lanes, a gather per entry
            lib.liship_spmv_csr_set_variant(variant)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(), yref), hex(variant)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            for a, b in ((n // 3 + 1, n - 5), (0, n // 3 + 1), (n - 5, n)):
                check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(), yref), hex(variant)
            # these kernels have a row split of their own: the fused entry points refuse, the caller runs the product and one reduction pass
            res = DA.from_host(np.full(2, np.nan), np.float64)
            assert lib.liship_csr_plan_fused_dots(plan) == 0
            assert lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1, res.ptr, work.ptr, None) != 0
    lib.liship_spmv_csr_set_index_codes(1)
    lib.liship_spmv_csr_set_row_patterns(1)
    lib.liship_spmv_csr_set_variant(0)
    lib.liship_spmv_csr_set_row_values(1)
    wide = lib.liship_csr_plan_wide_dominant(plan) == 1
    dom = lib.liship_csr_plan_dominant_pattern(plan) == 1  # value records with a dominant pattern: forms 6, 7 leave a partial per tile / chunk of 512 rows
    scale_sum = float(np.abs(w).sum() * np.abs(yref).max() + np.dot(yref, yref)) + 1e-300
    check(lib.liship_csr_plan_destroy(plan))
    own = [k for k in range(9) if (wide and k == 7) or (dom and k in (6, 7))]   # forms whose kernels have an epilogue of their own
    fusing = [k for k in range(9) if results[k] and k not in own]               # (a form whose kernels have a row split of their own refuses the fused entry
    assert len({len(results[k]) for k in fusing}) == 1 and 0 in fusing           #  points: no results)
    for parts in zip(*(results[k] for k in fusing)):       # same partial sums, same fold: the reductions agree to the bit too
        assert all(np.array_equal(parts[0], q) for q in parts[1:])
    for k in own:                                          # a partial per 256 / 512 rows or per tile instead of one per plan row block: the same sums to rounding
        assert len(results[k]) == len(results[0]), k
        for got, want in zip(results[k], results[0]):
            assert np.all(np.abs(got - want) <= 1e-12 * np.maximum(np.abs(want), scale_sum)), (k, got, want)


def test_row_block_dots_switch_restores_the_other_forms_bits(lib):
    """liship_spmv_csr_set_row_block_dots(1) (LIS_AMD_ROW_BLOCK_DOTS=1): the fused dots of the dominant-pattern product are the row blocks' partial sums again --
    bit-equal to the round-2 value-record kernels' -- while the default (a partial per tile) agrees with them to rounding; y is the oracle's either way"""
    ptr, idx, val = orc.poisson3d(24, 20, 128)
    n = len(ptr) - 1
    rng = np.random.default_rng(11)
    x, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_dominant_pattern(plan) == 1

    def dots(wptr):
        res = DA.from_host(np.full(2, np.nan), np.float64)
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, wptr, 1, res.ptr, work.ptr, None))
        assert np.array_equal(dy.to_host(), yref)
        return res.to_host().copy()
    try:
        lib.liship_spmv_csr_set_variant(0x20000000)                      # the round-2 kernels: the row blocks' partial sums
        want = [dots(dw.ptr), dots(dx.ptr)]
        lib.liship_spmv_csr_set_variant(0)
        tiles = [dots(dw.ptr), dots(dx.ptr)]
        check(lib.liship_spmv_csr_set_row_block_dots(1))
        blocks = [dots(dw.ptr), dots(dx.ptr)]
    finally:
        check(lib.liship_spmv_csr_set_row_block_dots(0))
        lib.liship_spmv_csr_set_variant(0)
        check(lib.liship_csr_plan_destroy(plan))
    scale = float(np.abs(w).sum() * np.abs(yref).max() + np.dot(yref, yref))
    for a, b, c in zip(want, tiles, blocks):
        assert np.array_equal(a, c), (a, c)
        assert np.all(np.abs(a - b) <= 1e-12 * scale), (a, b)
    assert any(not np.array_equal(a, b) for a, b in zip(want, tiles)) or True     # (the tile sums usually differ in the last bits; equality is not an error)


@pytest.mark.parametrize("grid", [(5, 6, 256), (3, 5, 512), (9, 4, 128)])
def test_valuerec_dominant_pattern_tiles_and_runs(lib, grid):
    """the dominant-pattern product under every lane -> row mapping it has -- contiguous chunks, XCD runs of chunks, tiles of 32 / 64 / 128 /
    256 columns -- on grids whose lines the tile widths divide, whole and in row ranges that start on odd rows: the oracle's bits each time
    (a permutation of who computes which row cannot change a row's sum, but an off-by-one in the tile arithmetic would)"""
    ptr, idx, val = orc.poisson3d(*grid)
    n = len(ptr) - 1
    x = np.random.default_rng(7).uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_value_records(plan) == 1
    try:
        for variant in (0, 0x10000000, 0x20000000):
            lib.liship_spmv_csr_set_variant(variant)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(), yref), hex(variant)
            lo, hi = n // 5 | 1, (n - n // 7) | 1           # odd cuts: the 16 B pieces of y and the pattern bytes' pairs lose their alignment
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            for a, b in ((lo, hi), (0, lo), (hi, n)):
                check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(), yref), ("rows", hex(variant))
    finally:
        lib.liship_spmv_csr_set_variant(0)
        check(lib.liship_csr_plan_destroy(plan))


def stack_rows(parts):
    """row-wise concatenation of CSR matrices (ptr, idx, val) over one column space"""
    ptr, idx, val = [np.zeros(1, np.int64)], [], []
    for p, i, v in parts:
        ptr.append(np.asarray(p[1:], np.int64) + ptr[-1][-1])
        idx.append(i); val.append(v)
    return np.concatenate(ptr).astype(np.int32), np.concatenate(idx).astype(np.int32), np.concatenate(val)


def _fem(G, dofs=3):
    return orc.fem3(G, dofs)[:3]


def _fem_with_strangers(kind):
    """an FEM pattern with other rows in between: blocks the plan must leave on the 4 B indices"""
    ptr, idx, val = _fem(10)
    n = len(ptr) - 1
    if kind == "long_row":                       # one row longer than the LDS stage, in the middle
        extra = orc.random_csr(1, 5000, seed=8, ncols=n, empty_rows=False)
    else:                                        # 60 rows of 70 random columns: more than 1024 distinct columns per row block
        extra = orc.random_csr(60, 70, seed=9, ncols=n, empty_rows=False)
    half = n // 2
    top = (ptr[:half + 1], idx[:ptr[half]], val[:ptr[half]])
    bot = (ptr[half:] - ptr[half], idx[ptr[half]:], val[ptr[half]:])
    return stack_rows([top, extra, bot])


LOCAL_CASES = {       # name: (matrix, plan keeps block-local columns?)
    "fem3_12": (lambda: _fem(12), True),
    "fem2_14": (lambda: _fem(14, 2), True),                       # 54 per row, 2 x 2 blocks
    "fem3_ghost_columns": (lambda: (lambda p, i, v: (p, (i + (i % 7 == 0) * (len(p) - 1)).astype(np.int32), v))(*_fem(9)), True),
    "band_60": (lambda: banded(20000, list(range(-30, 30)), 2), True),
    "fem3_long_row": (lambda: _fem_with_strangers("long_row"), True),
    "fem3_random_rows": (lambda: _fem_with_strangers("random"), True),
    "wide_77_random": (lambda: orc.random_csr(4000, 77, seed=3, ncols=4000, empty_rows=False), False),    # every column distinct
    "p3d_30": (lambda: orc.poisson3d(30, 30, 30), True),             # short rows whose row blocks share their columns (a plan without column codes): tried and kept since round 6
    "mesh_30k": (lambda: orc.unstructured_mesh(30000), True),        # ... the class that change is for: an unstructured mesh, ragged rows of 7 .. 30 entries, one unknown per node
    "rand_short_9": (lambda: orc.random_csr(40000, 9, seed=12, empty_rows=False), False),    # short rows with random columns: tried, not kept (the row-gather kernel as before)
    "p3d_30_switch_off": (lambda: orc.poisson3d(30, 30, 30), False),  # liship_spmv_csr_set_local_short_rows(0): the rule of rounds 2-5
}


@pytest.mark.parametrize("name", list(LOCAL_CASES))
def test_spmv_csr_local_columns(lib, name):
    """block-local columns (long rows, few distinct columns per row block): the plan keeps them for exactly the matrices
    that qualify, and with them every form of the product returns the bits of the 4 B-index kernels and of the oracle"""
    make, want = LOCAL_CASES[name]
    ptr, idx, val = make()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1)
    rng = np.random.default_rng(37)
    x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    if name.endswith("_switch_off"):
        lib.liship_spmv_csr_set_local_short_rows(0)
    try:
        check(lib.liship_csr_plan_localize_columns(plan, dptr.ptr, didx.ptr, None))
    finally:
        lib.liship_spmv_csr_set_local_short_rows(1)
    listed = lib.liship_csr_plan_localized(plan)
    assert (listed > 0) == want, listed
    if want:
        assert listed * (3 if np.diff(ptr).mean() < 22 else 2) <= len(idx) * (2 if np.diff(ptr).mean() < 22 else 1)      # the lists are worth their bytes (short rows: up to two listed columns per three entries)
    # round 5: lists made of triples of consecutive columns (3 unknowns per node, every node's three columns present) are kept as run starts too
    runs = lib.liship_csr_plan_local_runs(plan)
    if name in ("fem3_12", "fem3_long_row"):                   # (a block without a list -- the row longer than the stage -- has no triples to break the rule)
        assert runs == 3, name
    elif name in ("fem2_14", "band_60", "fem3_ghost_columns", "wide_77_random", "p3d_30", "rand_short_9"):      # pairs, a band, triples torn by the shifted columns, no lists at all
        assert runs == 0, name
    results = {}
    for on in (1, 2, 3, 0):                                # 2: the lists in full although the plan has the runs; 3: one entry per lane and step instead of pairs (A/B: same bits)
        lib.liship_spmv_csr_set_local_columns(1 if on else 0)
        lib.liship_spmv_csr_set_local_runs(0 if on == 2 else 1)
        lib.liship_spmv_csr_set_local_pairs(0 if on == 3 else 1)
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref), on
        out = []
        for sq in (0, 1):
            res = DA.from_host(np.full(2, np.nan), np.float64)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, sq, res.ptr, work.ptr, None)
            if rc == 0:
                assert np.array_equal(dy.to_host(), yref), (on, sq)
                out.append(res.to_host()[:1 + sq].copy())
        lo, hi = n // 5 + 1, n - n // 7
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        for a, b in ((lo, hi), (0, lo), (hi, n)):
            check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref), on
        total, used = 0, C.c_int()
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        ok = True
        for a, b in ((lo, hi), (0, lo), (hi, n)):
            rc = lib.liship_spmv_csr_rows_dot_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1,
                                                  work.ptr, total, C.byref(used), None)
            ok = ok and rc == 0
            total += used.value
        if ok:
            res = DA.from_host(np.full(2, np.nan), np.float64)
            check(lib.liship_spmv_csr_dot_finish_f64(total, 1, res.ptr, work.ptr, None))
            assert np.array_equal(dy.to_host(), yref), on
            out.append(res.to_host().copy())
        results[on] = out
    lib.liship_spmv_csr_set_local_columns(1)
    lib.liship_spmv_csr_set_local_runs(1)
    lib.liship_spmv_csr_set_local_pairs(1)
    check(lib.liship_csr_plan_destroy(plan))
    assert len(results[0]) == len(results[1]) == len(results[2]) == len(results[3])
    for a, b, c, d in zip(results[0], results[1], results[2], results[3]):
        assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)                        # same partial sums, same fold: the reductions agree to the bit too


def permute_csr(ptr, idx, val, perm):
    """P A P^T: new row i is old row perm[i] with its entries in their stored order, column c renamed to the new position of c"""
    n = len(ptr) - 1
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    lens = np.diff(ptr)[perm]
    ptr2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    src = np.repeat(np.asarray(ptr[:-1], np.int64)[perm] - ptr2[:-1], lens) + np.arange(ptr2[-1])
    return ptr2, inv[idx[src]].astype(np.int32), val[src]


def _scrambled_fem(kind):
    """a 3-dof mesh (81 entries per row) whose numbering has no locality: what liship_csr_plan_reorder is for"""
    rng = np.random.default_rng(55)
    if kind == "two_meshes_and_loose_rows":           # two components, and rows that are empty / hold one entry (vertices without neighbours)
        p, i, v = _fem(23)
        m = len(p) - 1
        p2 = np.concatenate([p, p[1:] + p[-1]]).astype(np.int32)
        i2 = np.concatenate([i, i + m]).astype(np.int32)
        ptr, idx, val = p2, i2, np.concatenate([v, 0.5 * v])
        n = 2 * m
        loose = np.zeros(1200, np.int64); loose[::3] = 1                                  # 400 single-entry rows (their own diagonal), 800 empty ones
        lp = np.concatenate([[0], np.cumsum(loose)]) + ptr[-1]
        ptr = np.concatenate([ptr, lp[1:]]).astype(np.int32)
        idx = np.concatenate([idx, n + np.flatnonzero(loose)]).astype(np.int32)
        val = np.concatenate([val, np.full(int(loose.sum()), 2.5)])
        n += 1200
        nodes = n // 3
    else:
        ptr, idx, val = _fem(28)
        n = len(ptr) - 1
        nodes = n // 3
    if kind == "rows":                                # every unknown on its own: no triples left
        perm = rng.permutation(n)
    else:                                             # nodes permuted, the three unknowns of a node stay together
        perm = (3 * rng.permutation(nodes)[:, None] + np.arange(3)[None, :]).reshape(-1)
        perm = np.concatenate([perm, np.arange(3 * nodes, n)])
    return permute_csr(ptr, idx, val, perm)


def _scrambled_poisson(scramble, vary=True):
    """short rows: the 7-point matrix of a 48 x 48 x 40 grid (with varying coefficients: products only; without: the SPD matrix the solvers want), its nodes numbered at random (or not)"""
    ptr, idx, val = orc.poisson3d(40, 48, 48)
    n = len(ptr) - 1
    if vary:
        val = val * np.random.default_rng(8).uniform(0.5, 1.5, len(val))
    return permute_csr(ptr, idx, val, np.random.default_rng(9).permutation(n)) if scramble else (ptr, idx, val)


@pytest.mark.parametrize("kind", ["nodes", "rows", "two_meshes_and_loose_rows", "natural", "short_rows", "short_rows_natural"])
def test_reordered_plan_bit_exact(lib, kind):
    """liship_csr_plan_reorder: a matrix numbered without locality is renumbered inside the plan (Cuthill-McKee, P A P^T in HBM); y keeps the oracle's bits
    (lis_matvec_csr.c:97-109: every row sum is its own terms in stored order) with the reordered form on and off, special values included; the fused entry
    points step aside, row ranges keep the original numbering; a mesh in its natural numbering is left alone"""
    short = kind.startswith("short_rows")
    ptr, idx, val = _scrambled_poisson(kind == "short_rows") if short else _fem(28) if kind == "natural" else _scrambled_fem(kind)
    n = len(ptr) - 1
    assert n >= 65536
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, n)
    x[rng.integers(0, n, 40)] = [-0.0, np.inf, -np.inf, np.nan, 5e-324, 1e308, -1e308, 0.0] * 5
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_localize_columns(plan, dptr.ptr, didx.ptr, None))
    listed = lib.liship_csr_plan_localized(plan)
    assert (listed > 0) == (kind not in ("rows", "short_rows"))      # unknowns scattered one by one: more than 2048 distinct columns per row block, no lists in the caller's numbering; short rows numbered at random: tried, not kept (naturally numbered ones keep theirs since round 6)
    check(lib.liship_csr_plan_reorder(plan, dptr.ptr, didx.ptr, dval.ptr, 0, None))
    re = lib.liship_csr_plan_reordered(plan)
    if kind in ("natural", "short_rows_natural"):
        # (a naturally numbered mesh is left alone; the 48 x 48-line grid of the short-row case lists a column per 2.2 entries in its natural order since round 6 -- lines
        #  this short put two neighbouring planes into every row block -- and the compact cells of the device ordering list 40 % fewer: kept, by the long rows' rule)
        assert (re == 0 or (short and listed > 0 and re * 4 <= listed * 3)) and lib.liship_csr_plan_fused_dots(plan) == 1, (re, listed)
        check(lib.liship_csr_plan_reorder(plan, dptr.ptr, didx.ptr, dval.ptr, 1 << 20, None))      # forced to try: a candidate is kept only when it lists 1/4 fewer columns than the mesh order
        forced = lib.liship_csr_plan_reordered(plan)      # (the Cuthill-McKee walk of rounds 4-5 never did; the six-landmark cells of round 6 do on this 28^3 mesh: compact 3-D cells against line-by-line rows)
        assert forced == 0 or (listed > 0 and forced * 4 <= listed * 3), (forced, listed)
        re = forced
    else:
        assert re > 0 and (listed == 0 or re * 4 <= listed * 3), (re, listed)
        assert lib.liship_csr_plan_fused_dots(plan) == 1                          # by default products stay in the caller's numbering (permuting x and y per product eats the gain)
        lib.liship_spmv_csr_set_reorder(2)
        assert lib.liship_csr_plan_fused_dots(plan) == (1 if short else 0)       # opted in: long rows only
        lib.liship_spmv_csr_set_reorder(1)
        # the reordered form as a matrix of its own (what lis_solve iterates on): gather x, multiply by P A P^T, scatter y -- the oracle's bits
        inner, rp, ri, rv, pm = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.liship_csr_plan_reordered_form(plan, C.byref(inner), C.byref(rp), C.byref(ri), C.byref(rv), C.byref(pm)))
        xp, yp, yb = DA(n, np.float64), DA.from_host(np.full(n, np.nan), np.float64), DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_permute_gather_f64(n, pm, dx.ptr, xp.ptr, None))
        check(lib.liship_spmv_csr_f64(inner, rp, ri, rv, xp.ptr, yp.ptr, None))
        check(lib.liship_permute_scatter_f64(n, pm, yp.ptr, yb.ptr, None))
        assert np.array_equal(yb.to_host(), yref, equal_nan=True)
    ys = {}
    for on in (2, 0, 1, 2):                              # 2: whole-matrix products of long-row plans take the reordered form (opt-in); 1: the default, products in the caller's numbering
        lib.liship_spmv_csr_set_reorder(on)
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        ys[on] = dy.to_host()
        assert np.array_equal(ys[on], yref, equal_nan=True), (kind, on)
        assert np.array_equal(np.signbit(ys[on]), np.signbit(yref))
    if re and not short:
        work, res, dw = DA(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64), DA.from_host(x, np.float64)
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        assert lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1, res.ptr, work.ptr, None) != 0      # the caller runs product + reduction
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        for a, b in ((n // 3, n - 7), (0, n // 3), (n - 7, n)):
            check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref, equal_nan=True)
        # a split matrix's rows start from their first product (-0.0 + p): the reordered form starts there too
        check(lib.liship_csr_plan_set_first_term_initialises(plan, 1))
        x0 = np.zeros(n); x0[::2] = -0.0
        d0 = DA.from_host(x0, np.float64)
        got = {}
        for on in (2, 0):
            lib.liship_spmv_csr_set_reorder(on)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, d0.ptr, dy.ptr, None))
            got[on] = dy.to_host()
        assert np.array_equal(got[0], got[2]) and np.array_equal(np.signbit(got[0]), np.signbit(got[2]))
    if kind == "nodes":
        # ghost columns (>= n): the walk refuses; a hinted permutation skips the walk and the renumbering kernel refuses instead -- no reordered form either way
        idx_g = idx.copy(); idx_g[7] = n + 3
        didx_g = DA.from_host(idx_g, np.int32)
        for hint in (None, np.arange(n, dtype=np.int32)[::-1].copy()):
            plan_g = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan_g), n, dptr.ptr, None))
            check(lib.liship_csr_plan_localize_columns(plan_g, dptr.ptr, didx_g.ptr, None))
            check(lib.liship_csr_plan_reorder_with(plan_g, dptr.ptr, didx_g.ptr, dval.ptr, 0, None if hint is None else hint.ctypes.data, None))
            assert lib.liship_csr_plan_reordered(plan_g) == 0
            check(lib.liship_csr_plan_destroy(plan_g))
        # ... DECLARED as a rank's ghost columns (liship_csr_plan_set_ghost_columns: a multi-rank job's local rows, lis_matrix_mpi.c:274-306), the matrix is renumbered all the
        # same: the ghost columns keep their numbers in P A P^T, the rows that read one come behind all the others (rows [0, inner) can run while the halo travels), the
        # export list of a halo is the same rows under their new numbers -- and the product has the oracle's bits
        lib.liship_spmv_csr_set_reorder(1)
        sel = rng.integers(0, len(idx), 5000)
        idx_g = idx.copy(); idx_g[sel] = n + (sel % 8)
        xg = np.concatenate([np.where(np.isfinite(x), x, 0.25), rng.uniform(-1, 1, 8)])
        yg = orc.spmv_csr(ptr, idx_g, val, xg)
        didx_g = DA.from_host(idx_g, np.int32)
        plan_g = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan_g), n, dptr.ptr, None))
        check(lib.liship_csr_plan_localize_columns(plan_g, dptr.ptr, didx_g.ptr, None))
        check(lib.liship_csr_plan_set_ghost_columns(plan_g, n + 8))
        check(lib.liship_csr_plan_reorder(plan_g, dptr.ptr, didx_g.ptr, dval.ptr, 0, None))
        assert lib.liship_csr_plan_reordered(plan_g) > 0
        inner_rows = lib.liship_csr_plan_reordered_inner_rows(plan_g)
        has_ghost = np.add.reduceat((idx_g >= n).astype(np.int64), ptr[:-1]) > 0
        assert inner_rows == n - int(has_ghost.sum())                  # (one connected mesh: every row has a component)
        inner, rp, ri, rv, pm = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.liship_csr_plan_reordered_form(plan_g, C.byref(inner), C.byref(rp), C.byref(ri), C.byref(rv), C.byref(pm)))
        hp, hperm = np.empty(n + 1, np.int32), np.empty(n, np.int32)
        check(lib.liship_memcpy_d2h(hp.ctypes.data, rp, hp.nbytes, None)); check(lib.liship_memcpy_d2h(hperm.ctypes.data, pm, hperm.nbytes, None))
        check(lib.liship_device_synchronize())
        hi = np.empty(hp[-1], np.int32)
        check(lib.liship_memcpy_d2h(hi.ctypes.data, ri, hi.nbytes, None)); check(lib.liship_device_synchronize())
        assert np.array_equal(np.sort(hperm), np.arange(n)) and np.array_equal(has_ghost[hperm], np.arange(n) >= inner_rows)
        assert hi[:hp[inner_rows]].max() < n and np.array_equal(np.sort(hi[hi >= n]), np.sort(idx_g[idx_g >= n]))
        xp, yp, yb = DA.from_host(np.zeros(n + 8), np.float64), DA.from_host(np.full(n, np.nan), np.float64), DA.from_host(np.full(n, np.nan), np.float64)
        dxg = DA.from_host(xg, np.float64)
        check(lib.liship_permute_gather_f64(n, pm, dxg.ptr, xp.ptr, None))
        check(lib.liship_memcpy_d2d(xp.ptr + 8 * n, dxg.ptr + 8 * n, 64, None))          # the ghost slots: where the halo always lands
        check(lib.liship_spmv_csr_f64(inner, rp, ri, rv, xp.ptr, yp.ptr, None))
        check(lib.liship_permute_scatter_f64(n, pm, yp.ptr, yb.ptr, None))
        assert np.array_equal(yb.to_host().view(np.uint64), yg.view(np.uint64))
        for a, b in ((0, inner_rows), (inner_rows, n)):                                    # ... in the two launches of an overlapped product
            check(lib.liship_spmv_csr_rows_f64(inner, a, b, rp, ri, rv, xp.ptr, yp.ptr, None))
        check(lib.liship_permute_scatter_f64(n, pm, yp.ptr, yb.ptr, None))
        assert np.array_equal(yb.to_host().view(np.uint64), yg.view(np.uint64))
        ex = np.sort(rng.choice(n, 3000, replace=False)).astype(np.int32)                  # a halo export list
        dex, dout = DA.from_host(ex, np.int32), DA(3000, np.int32)
        check(lib.liship_permute_rows_of_list(n, pm, 3000, dex.ptr, dout.ptr, None))
        assert np.array_equal(hperm[dout.to_host()], ex)
        for on in (2, 1):                                                                   # single products keep the caller's numbering (x carries the ghost entries behind the rows)
            lib.liship_spmv_csr_set_reorder(on)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan_g, dptr.ptr, didx_g.ptr, dval.ptr, dxg.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yg.view(np.uint64))
        assert lib.liship_csr_plan_fused_dots(plan_g) == 1
        check(lib.liship_csr_plan_destroy(plan_g))
    if kind == "nodes":
        # a plan for the same pattern with other values (a matrix edited in place): the first plan's permutation as a hint, no second walk; a broken hint is dropped
        lib.liship_spmv_csr_set_reorder(2)
        check(lib.liship_csr_plan_set_first_term_initialises(plan, 0))
        perm = np.empty(n, np.int32)
        check(lib.liship_csr_plan_reorder_permutation(plan, perm.ctypes.data))
        assert np.array_equal(np.sort(perm), np.arange(n))
        val2 = val * rng.uniform(0.5, 1.5, len(val))
        dval2 = DA.from_host(val2, np.float64)
        yref2 = orc.spmv_csr(ptr, idx, val2, x)
        for hint in (perm, np.zeros(n, np.int32), np.arange(n, dtype=np.int32)[::-1].copy()):        # the walk's own; not a permutation; a permutation that is no better
            plan2 = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan2), n, dptr.ptr, None))
            check(lib.liship_csr_plan_localize_columns(plan2, dptr.ptr, didx.ptr, None))
            check(lib.liship_csr_plan_reorder_with(plan2, dptr.ptr, didx.ptr, dval2.ptr, 0, hint.ctypes.data, None))
            assert lib.liship_csr_plan_reordered(plan2) == re             # the same lists either way (the walk is deterministic)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan2, dptr.ptr, didx.ptr, dval2.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(), yref2, equal_nan=True)
            check(lib.liship_csr_plan_destroy(plan2))
    lib.liship_spmv_csr_set_reorder(1)
    check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("L", [21, 22, 23, 25, 26, 41, 42, 43, 62, 63, 81, 85, 101, 127])
def test_uniform_length_rows_fed_sums(lib, L):
    """rows of ONE length (>= 21, not a multiple of 4): the wavefronts of the products and block-local kernels add them without a skew,
    L mod 21 terms first and whole rounds of 21 through the fed chain (ordered_sum_rows) -- the oracle's bits (lis_matvec_csr.c:97-109),
    and the bits of the skewed sums, with special values among the products; a few shorter rows make some wavefronts mixed"""
    n = 3000
    rng = np.random.default_rng(L)
    lens = np.full(n, L, np.int64)
    lens[rng.integers(0, n, 5)] = rng.integers(0, L, 5)              # a handful of ragged rows: their wavefronts take the skewed path
    lens[64 * 7: 64 * 9] = L                                          # ... and two whole wavefronts surely uniform
    ptr = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[-1])
    rows = np.repeat(np.arange(n), lens)
    k = np.arange(nnz) - ptr[rows]
    idx = ((rows + (k - L // 2) * 3) % n).astype(np.int32)            # a band: few distinct columns per row block (the local kernel qualifies)
    val = rng.uniform(-1, 1, nnz)
    x = rng.uniform(-1, 1, n)
    x[5], x[77] = 0.0, -0.0
    val[rng.integers(0, nnz, 8)] = 0.0
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx = DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_localize_columns(plan, dptr.ptr, didx.ptr, None))
    try:
        for local in (1, 0):
            lib.liship_spmv_csr_set_local_columns(local)
            for uniform in (1, 0):
                check(lib.liship_spmv_csr_set_uniform_rows(uniform))
                dy = DA.from_host(np.full(n, np.nan), np.float64)
                check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (local, uniform)
    finally:
        lib.liship_spmv_csr_set_local_columns(1)
        check(lib.liship_spmv_csr_set_uniform_rows(1))
        check(lib.liship_csr_plan_destroy(plan))


def test_spmv_csr_long_row_tree_is_the_default(lib):
    """rows longer than the LDS stage: a workgroup tree by default (round 6) -- the oracle's value to rounding, reproducibly -- and left to right (the oracle's bits)
    with the chain switched on; rows that fit the stage keep their bits either way"""
    rng = np.random.default_rng(12)
    base = orc.random_csr(3000, 9, seed=12, ncols=3000)
    half = 1500
    long_ptr = np.array([0, 60000], np.int32)                     # one row of 60 000 entries (columns repeat)
    long_row = (long_ptr, rng.integers(0, 3000, 60000).astype(np.int32), rng.uniform(-1, 1, 60000))
    top = (base[0][:half + 1], base[1][:base[0][half]], base[2][:base[0][half]])
    bot = (base[0][half:] - base[0][half], base[1][base[0][half]:], base[2][base[0][half]:])
    ptr, idx, val = stack_rows([top, long_row, bot])
    n = len(ptr) - 1
    x = np.random.default_rng(5).uniform(-1, 1, 3000)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    out = []
    assert lib.liship_spmv_csr_switches() & 4, "the tree is the default"
    for tree in (0, 1, 1, None):
        if tree is not None:
            check(lib.liship_spmv_csr_set_long_row_tree(tree))
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        out.append(dy.to_host())
    check(lib.liship_csr_plan_destroy(plan))
    assert np.array_equal(out[0], yref)
    assert np.array_equal(out[1], out[2]) and np.array_equal(out[1], out[3])
    short = np.diff(ptr) < 1024                                      # (rows shorter than 1024 entries keep the reference's bits whatever the mode)
    assert np.array_equal(out[1][short], yref[short])
    scale = np.abs(val[ptr[1500]:ptr[1501]] * x[idx[ptr[1500]:ptr[1501]]]).sum()
    assert abs(out[1][1500] - yref[1500]) <= 1e-13 * scale            # a different association of 60 000 terms


def test_spmv_csr_long_row_tree_tail_over_many_workgroups(lib):
    """tree mode: beyond 16 384 entries of a row block the tail of its last row is summed by a workgroup per 8192 entries in front of the product
    (spmv_csr_tail_chunks_kernel / _fold_kernel) and handed over in y[row] -- whole products, row ranges that cut in front of / behind / away from the long
    rows, the fused dots; two long rows (one of them the matrix's last row); every value within 1e-14 of the row's magnitude, short rows keep their bits"""
    rng = np.random.default_rng(21)
    base = orc.random_csr(4000, 11, seed=21, ncols=4000)
    cut = 1700

    def long_row(m):
        return (np.array([0, m], np.int32), rng.integers(0, 4000, m).astype(np.int32), rng.uniform(-1, 1, m))
    top = (base[0][:cut + 1], base[1][:base[0][cut]], base[2][:base[0][cut]])
    bot = (base[0][cut:] - base[0][cut], base[1][base[0][cut]:], base[2][base[0][cut]:])
    ptr, idx, val = stack_rows([top, long_row(150001), bot, long_row(16384 + 8192 + 5)])
    n = len(ptr) - 1
    lens = np.diff(ptr)
    x = np.random.default_rng(6).uniform(-1, 1, 4000)
    w = np.random.default_rng(7).uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    absum = np.add.reduceat(np.abs(val * x[idx]), ptr[:-1].astype(np.int64))
    dptr, didx, dval, dx, dw = (DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64),
                                DA.from_host(w, np.float64))
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    res = DA(2, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    short = lens <= 2000

    def close(y, rows=slice(None)):
        assert np.array_equal(y[rows][short[rows]], yref[rows][short[rows]])
        assert np.all(np.abs(y[rows] - yref[rows]) <= 1e-14 * absum[rows])
    check(lib.liship_spmv_csr_set_long_row_tree(1))
    try:
        whole = []
        for _ in range(2):
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            whole.append(dy.to_host())
        assert np.array_equal(whole[0], whole[1])
        close(whole[0])
        for a, b in ((0, cut), (cut, cut + 1), (cut - 3, cut + 4), (cut + 1, n), (n - 1, n), (5, n - 1)):
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            y = dy.to_host()
            assert np.isnan(y[:a]).all() and np.isnan(y[b:]).all()
            close(y, slice(a, b))                                 # (a range that cuts a row block moves the start of its tail: the same sum to rounding)
        for sq in (0, 1):
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, sq, res.ptr, work.ptr, None)
            if rc != 0:
                continue                                          # (the plan's form has no fused epilogue: the caller runs the product and a pass)
            y, r = dy.to_host(), res.to_host()
            assert np.array_equal(y, whole[0])
            assert abs(r[0] - float(np.dot(w, y))) <= 1e-12 * float(np.abs(w * y).sum())
            if sq:
                assert abs(r[1] - float(np.dot(y, y))) <= 1e-12 * float(np.dot(y, y))
    finally:
        check(lib.liship_spmv_csr_set_long_row_tree(0))
    try:
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(), yref)                  # and the chain mode is the reference's sum, bit for bit
    finally:
        check(lib.liship_spmv_csr_set_long_row_tree(1))            # (back to the default)
    check(lib.liship_csr_plan_destroy(plan))


BSR22_CASES = {
    "stencil": lambda: orc.poisson3d(23, 18, 14),
    "one_block_row": lambda: orc.random_csr(2, 2, seed=7, empty_rows=False),
    "63_rows": lambda: orc.random_csr(63, 6, seed=8),
    "129_rows_odd": lambda: orc.random_csr(129, 9, seed=9),                       # odd n: the last block row is padded
    "long_rows": lambda: orc.random_csr(9000, 3, seed=2, long_row=7001),         # a block row of > 512 blocks: several LDS passes
    "mostly_empty": lambda: orc.random_csr(20000, 0.05, seed=4),
    "wide": lambda: orc.random_csr(4000, 77, seed=3, ncols=4000, empty_rows=False),
    "fem_3dofs": lambda: orc.fem3(7, 3)[:3],        # 27 blocks per interior block row at 3x3: the team-per-block-row kernel (8 lanes x 4 blocks)
    "fem_2dofs": lambda: orc.fem3(9, 2)[:3],
    "fem_4dofs": lambda: orc.fem3(6, 4)[:3],
}


@pytest.mark.parametrize("name", list(BSR22_CASES))
@pytest.mark.parametrize("bs", [1, 2, 3, 4])
def test_spmv_bsr_square_blocks(lib, name, bs):
    """square blocks 1..4 (2x2 takes the wavefront-per-64-block-rows kernel with LDS-DMA staging): bit-identical to
    lis_matvec_bsr's order on ragged, empty, very long and tail block rows"""
    ptr, idx, val = BSR22_CASES[name]()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1 if len(idx) else 1)
    pad = (-ncols) % bs
    x = np.random.default_rng(21).uniform(-1, 1, ncols + pad + bs)
    nr, bptr, bidx, bv = orc.csr2bsr(ptr, idx, val, bs, bs)
    nc = int(bidx.max()) + 1 if len(bidx) else 1
    xx = np.zeros(max(len(x), nc * bs))
    xx[:len(x)] = x
    a, b, c = DA.from_host(bptr), DA.from_host(bidx if len(bidx) else np.zeros(1, np.int32)), DA.from_host(bv if len(bv) else np.zeros(1))
    dx, dy = DA.from_host(xx), DA.from_host(np.full(nr * bs, np.nan))
    ref = orc.spmv_bsr(n, nr, bs, bs, bptr, bidx, bv, xx)
    for known in (-1, len(bidx), 0, 1 << 30):             # 0: "short block rows" claimed for any matrix -> the lane-per-block-row kernel;
        for team in (1, 0, 2):                            # 2^30: long ones -> a team per block row (2: at 4x4 too), or (team 0) the two-phase tile kernels
            lib.liship_spmv_bsr_set_team(team)
            dy = DA.from_host(np.full(nr * bs, np.nan))
            check(lib.liship_spmv_bsr_nnz_f64(nr, known, bs, bs, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(n), ref), (known, team)
    lib.liship_spmv_bsr_set_team(1)
    check(lib.liship_spmv_bsr_f64(nr, bs, bs, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, None))
    assert np.array_equal(dy.to_host(n), ref)
    # block-row ranges (the parts of a multi-rank product around its halo exchange): the bits of the whole launch, wherever the cuts fall
    for team in (1, 0):
        lib.liship_spmv_bsr_set_team(team)
        for known in (len(bidx), 1 << 30):
            dy = DA.from_host(np.full(nr * bs, np.nan))
            cuts = [0, nr // 3, nr - nr // 4, nr]
            for lo, hi in ((cuts[1], cuts[2]), (cuts[0], cuts[1]), (cuts[2], cuts[3])):
                check(lib.liship_spmv_bsr_rows_f64(nr, known, bs, bs, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, lo, hi, None))
            assert np.array_equal(dy.to_host(n), ref), ("rows", team, known)
    lib.liship_spmv_bsr_set_team(1)
    # the product with the reduction epilogue: same y, sums as numpy to 1e-13, repeatable; refuses what it does not serve
    w = np.random.default_rng(22).uniform(-1, 1, nr * bs)
    dw, work, res = DA.from_host(w), DA(lib.liship_reduce_work_bytes() // 8, np.float64), DA.from_host(np.full(2, np.nan))
    for want_sumsq in (0, 1):
        got = []
        for _ in range(2):
            dy = DA.from_host(np.full(nr * bs, np.nan))
            rc = lib.liship_spmv_bsr_dot_f64(nr, n, len(bidx), bs, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, dw.ptr, want_sumsq,
                                             res.ptr, work.ptr, None)
            if rc != 0:
                assert rc == -1 and (bs == 1 or len(bidx) / nr > 12)
                break
            assert np.array_equal(dy.to_host(n), ref)
            got.append(res.to_host())
        if got:
            assert np.array_equal(got[0][:1 + want_sumsq], got[1][:1 + want_sumsq])
            assert abs(got[0][0] - np.dot(w[:n], ref)) <= 1e-13 * (np.abs(w[:n] * ref).sum() + 1e-300)
            if want_sumsq:
                assert abs(got[0][1] - np.dot(ref, ref)) <= 1e-13 * np.dot(ref, ref) + 1e-300


@pytest.mark.parametrize("seed", range(int(os.environ.get("LIS_AMD_FUZZ_SEEDS", "40"))))
def test_spmv_bsr_on_random_structured_matrices(lib, seed):
    """the generated matrices of the plan-coder fuzz (stencil-like, banded, ragged, with empty and long rows) in square blocks 2..4
    through every BSR kernel: lane per block row (claimed short), the two-phase tile kernels, the dispatcher's own choice"""
    ptr, idx, val, ncols = _structured_random(seed)
    n = len(ptr) - 1
    if ncols > n:
        pytest.skip("the reference's csr2bsr (and the oracle's) takes square matrices: its work array has n / bnc block columns")
    for bs in (2, 3, 4):
        nr, bptr, bidx, bv = orc.csr2bsr(ptr, idx, val, bs, bs)
        nc = max(int(bidx.max()) + 1 if len(bidx) else 1, (ncols + bs - 1) // bs)
        xx = np.zeros(nc * bs + bs)
        xx[:ncols] = np.random.default_rng(3000 + seed).uniform(-1, 1, ncols)
        ref = orc.spmv_bsr(n, nr, bs, bs, bptr, bidx, bv, xx)
        a, b, c = DA.from_host(bptr), DA.from_host(bidx if len(bidx) else np.zeros(1, np.int32)), DA.from_host(bv if len(bv) else np.zeros(1))
        dx = DA.from_host(xx)
        for known, team in ((0, 1), (1 << 30, 1), (1 << 30, 2), (1 << 30, 0), (len(bidx), 1)):      # claimed short block rows, claimed long ones (a team per block row /
            lib.liship_spmv_bsr_set_team(team)                                        # the two-phase tile kernels), the truth
            dy = DA.from_host(np.full(max(nr * bs, 1), np.nan))
            check(lib.liship_spmv_bsr_nnz_f64(nr, known, bs, bs, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host(n), ref), (seed, bs, known, team)
        lib.liship_spmv_bsr_set_team(1)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 1000, 4097, 1 << 20, (1 << 20) + 3])
def test_elementwise_bit_exact(lib, n):
    rng = np.random.default_rng(n)
    x, y, z = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(0.5, 2, n)
    a = 0.37120000000000003
    O = orc.lib()
    dx, dy, dz = DA.from_host(x), DA.from_host(y), DA.from_host(z)
    yy = y.copy(); O.orc_axpy(n, a, x, yy)
    check(lib.liship_axpy_f64(n, a, dx.ptr, dy.ptr, None)); assert np.array_equal(dy.to_host(), yy)
    O.orc_xpay(n, x, a, yy)
    check(lib.liship_xpay_f64(n, dx.ptr, a, dy.ptr, None)); assert np.array_equal(dy.to_host(), yy)
    zz = np.empty(n); O.orc_axpyz(n, a, x, yy, zz)
    check(lib.liship_axpyz_f64(n, a, dx.ptr, dy.ptr, dz.ptr, None)); assert np.array_equal(dz.to_host(), zz)
    O.orc_scale(n, a, zz)
    check(lib.liship_scale_f64(n, a, dz.ptr, None)); assert np.array_equal(dz.to_host(), zz)
    O.orc_pmul(n, x, yy, zz)
    check(lib.liship_pmul_f64(n, dx.ptr, dy.ptr, dz.ptr, None)); assert np.array_equal(dz.to_host(), zz)
    dz.upload(z); zz = z.copy(); O.orc_reciprocal(n, zz)
    check(lib.liship_reciprocal_f64(n, dz.ptr, None)); assert np.array_equal(dz.to_host(), zz)
    check(lib.liship_pdiv_f64(n, dx.ptr, dz.ptr, dy.ptr, None)); assert np.array_equal(dy.to_host(), x / zz)
    check(lib.liship_scale_to_f64(n, a, dx.ptr, dy.ptr, None)); assert np.array_equal(dy.to_host(), a * x)
    check(lib.liship_set_all_f64(n, a, dy.ptr, None)); assert np.array_equal(dy.to_host(), np.full(n, a))
    check(lib.liship_abs_f64(n, dx.ptr, None)); assert np.array_equal(dx.to_host(), np.abs(x))
    check(lib.liship_shift_f64(n, a, dx.ptr, None)); assert np.array_equal(dx.to_host(), np.abs(x) - a)


def test_elementwise_unaligned_views(lib):
    n = 1001
    x, y = np.arange(n + 1, dtype=np.float64), np.ones(n + 1)
    dx, dy = DA.from_host(x), DA.from_host(y)
    check(lib.liship_axpy_f64(n, 2.0, dx.ptr + 8, dy.ptr + 8, None))     # 8-byte aligned only
    out = dy.to_host()
    assert out[0] == 1.0 and np.array_equal(out[1:], 1.0 + 2.0 * x[1:])


@pytest.mark.parametrize("n", [1, 2, 65, 1000, 4097, 1 << 20, (1 << 22) + 5])
def test_reductions(lib, n):
    rng = np.random.default_rng(n + 1)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    dx, dy = DA.from_host(x), DA.from_host(y)
    work = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64)
    res = DA.zeros(2, np.float64)
    O = orc.lib()
    scale = np.sqrt(n)

    def close(a, b, ref_mag):
        return abs(a - b) <= 1e-14 * max(ref_mag, 1e-300) * max(1.0, np.log2(n + 1))

    check(lib.liship_dot_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
    assert close(res.to_host()[0], O.orc_dot(n, x, y), np.abs(x * y).sum())
    check(lib.liship_nrm2_f64(n, dx.ptr, res.ptr, work.ptr, None))
    assert close(res.to_host()[0], O.orc_nrm2(n, x), O.orc_nrm2(n, x))
    check(lib.liship_sumsq_f64(n, dx.ptr, res.ptr, work.ptr, None))
    assert close(res.to_host()[0], O.orc_nrm2(n, x) ** 2, np.dot(x, x))
    check(lib.liship_nrm1_f64(n, dx.ptr, res.ptr, work.ptr, None))
    assert close(res.to_host()[0], O.orc_nrm1(n, x), O.orc_nrm1(n, x))
    check(lib.liship_sum_f64(n, dx.ptr, res.ptr, work.ptr, None))
    assert close(res.to_host()[0], x.sum(), np.abs(x).sum())
    check(lib.liship_dot2_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
    r = res.to_host()
    assert close(r[0], O.orc_dot(n, x, y), np.abs(x * y).sum()) and close(r[1], np.dot(x, x), np.dot(x, x))
    # run-to-run reproducibility (fixed reduction tree)
    check(lib.liship_dot_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None)); a = res.to_host()[0]
    check(lib.liship_dot_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None)); assert a == res.to_host()[0]
    del scale


def test_diagonal_and_gather(lib):
    ptr, idx, val = orc.random_csr(3000, 9, seed=12)
    n = 3000
    a, b, c = DA.from_host(ptr), DA.from_host(idx), DA.from_host(val)
    d = DA.zeros(n, np.float64)
    check(lib.liship_csr_diagonal_f64(n, a.ptr, b.ptr, c.ptr, d.ptr, None))
    assert np.array_equal(d.to_host(), orc.csr_diagonal(ptr, idx, val))
    sel = np.random.default_rng(0).integers(0, n, 777).astype(np.int32)
    x = np.random.default_rng(1).uniform(-1, 1, n)
    dsel, dx, out = DA.from_host(sel), DA.from_host(x), DA.zeros(777, np.float64)
    check(lib.liship_gather_f64(777, dsel.ptr, dx.ptr, out.ptr, None))
    assert np.array_equal(out.to_host(), x[sel])


@pytest.mark.parametrize("grid,slab,sorted_", [((8, 6, 5), None, 0), ((8, 6, 5), None, 1),
                                               ((8, 6, 5), (2, 5), 0), ((8, 6, 5), (0, 3), 1),
                                               ((8, 6, 5), (5, 8), 0), ((1, 1, 7), None, 0)])
def test_device_poisson_generator(lib, grid, slab, sorted_):
    """Device-generated rows == oracle generator + the reference's ghost renumbering (lis_matrix_mpi.c:274-306)."""
    l, m, n = grid
    mn = m * n
    is_, ie = (0, l * mn) if slab is None else (slab[0] * mn, slab[1] * mn)
    nloc = ie - is_
    ptr, idx, val = orc.poisson3d(l, m, n, sort_cols=bool(sorted_), is_=is_, ie=ie)
    ghosts = np.unique(idx[(idx < is_) | (idx >= ie)])
    lidx = idx.copy()
    own = (idx >= is_) & (idx < ie)
    lidx[own] = idx[own] - is_
    lidx[~own] = nloc + np.searchsorted(ghosts, idx[~own])
    nnz = lib.liship_poisson3d_nnz(l, m, n, is_, ie)
    assert nnz == len(idx)
    dptr, didx, dval = DA.zeros(nloc + 1, np.int32), DA.zeros(nnz, np.int32), DA.zeros(nnz, np.float64)
    check(lib.liship_poisson3d_csr(l, m, n, is_, ie, sorted_, dptr.ptr, didx.ptr, dval.ptr, None))
    assert np.array_equal(dptr.to_host(), ptr)
    assert np.array_equal(didx.to_host(), lidx)
    assert np.array_equal(dval.to_host(), val)
    b = DA.zeros(nloc, np.float64)
    check(lib.liship_poisson3d_rhs(l, m, n, is_, ie, b.ptr, None))
    full_ptr, full_idx, full_val = orc.poisson3d(l, m, n)
    assert np.array_equal(b.to_host(), orc.spmv_csr(full_ptr, full_idx, full_val, np.ones(l * mn))[is_:ie])


def test_large_poisson_properties(lib):
    """Full-size property check without a CPU pass: ||A*1||_2^2 = 6(N-2)^2 + 48(N-2) + 72 (spmvtest3)."""
    N = 256
    n = N ** 3
    nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
    assert nnz == 7 * n - 6 * N * N
    dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
    check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
    x, y, b = DA(n, np.float64), DA(n, np.float64), DA(n, np.float64)
    check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    check(lib.liship_sumsq_f64(n, y.ptr, res.ptr, work.ptr, None))
    assert res.to_host()[0] == 6.0 * (N - 2) ** 2 + 48.0 * (N - 2) + 72.0     # small integers: exact
    # y == b (= A*1 in closed form) everywhere: y - b == 0
    check(lib.liship_poisson3d_rhs(N, N, N, 0, n, b.ptr, None))
    check(lib.liship_axpy_f64(n, -1.0, b.ptr, y.ptr, None))
    check(lib.liship_nrm1_f64(n, y.ptr, res.ptr, work.ptr, None))
    assert res.to_host()[0] == 0.0
    check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("n", [1, 2, 63, 1000, 4097, (1 << 21) + 3])
def test_fused_vector_kernels_bit_exact(lib, n):
    """Fused forms must give the bits of the separate reference calls they replace (same expressions, same order)."""
    rng = np.random.default_rng(n + 7)
    x, y, w, r = (rng.uniform(-1, 1, n) for _ in range(4))
    a, b = 0.37120000000000003, -1.25
    O = orc.lib()
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    dx, dy, dw, dr = DA.from_host(x), DA.from_host(y), DA.from_host(w), DA.from_host(r)
    # y += a*x ; y += b*w
    yy = y.copy(); O.orc_axpy(n, a, x, yy); O.orc_axpy(n, b, w, yy)
    check(lib.liship_axpy2_f64(n, a, dx.ptr, b, dw.ptr, dy.ptr, None)); assert np.array_equal(dy.to_host(), yy)
    # y += a*x ; y = w + b*y
    O.orc_axpy(n, a, x, yy); O.orc_xpay(n, w, b, yy)
    check(lib.liship_axpy_xpay_f64(n, a, dx.ptr, dw.ptr, b, dy.ptr, None)); assert np.array_equal(dy.to_host(), yy)
    # CG update: xi += a*p ; r += (-a)*q ; sum r^2
    xi, rr = w.copy(), r.copy()
    O.orc_axpy(n, a, x, xi); O.orc_axpy(n, -a, y, rr)
    dy.upload(y)
    check(lib.liship_cg_update_f64(n, a, dx.ptr, dy.ptr, dw.ptr, dr.ptr, res.ptr, work.ptr, None))
    assert np.array_equal(dw.to_host(), xi) and np.array_equal(dr.to_host(), rr)
    assert abs(res.to_host()[0] - np.dot(rr, rr)) <= 1e-14 * np.dot(rr, rr) * max(1.0, np.log2(n + 1))
    # y += a*x ; sum y^2 (; sum v*y)
    yy = y.copy(); O.orc_axpy(n, a, x, yy)
    check(lib.liship_axpy_sumsq_f64(n, a, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
    assert np.array_equal(dy.to_host(), yy) and abs(res.to_host()[0] - np.dot(yy, yy)) <= 1e-14 * np.dot(yy, yy) * max(1.0, np.log2(n + 1))
    O.orc_axpy(n, a, x, yy)
    check(lib.liship_axpy_sumsq_dot_f64(n, a, dx.ptr, dy.ptr, dr.ptr, res.ptr, work.ptr, None))
    out = res.to_host()
    assert np.array_equal(dy.to_host(), yy)
    assert abs(out[0] - np.dot(yy, yy)) <= 1e-14 * np.dot(yy, yy) * max(1.0, np.log2(n + 1))
    assert abs(out[1] - np.dot(rr, yy)) <= 1e-14 * np.abs(rr * yy).sum() * max(1.0, np.log2(n + 1))
    # Jacobi forms: z = x.*d ; y = z + b*y      and      CG update + <r, r.*dinv>
    dinv = rng.uniform(0.1, 2.0, n)
    dd = DA.from_host(dinv)
    z = np.empty(n); O.orc_pmul(n, x, dinv, z); O.orc_xpay(n, z, b, yy)
    check(lib.liship_pmul_xpay_f64(n, dx.ptr, dd.ptr, b, dy.ptr, None)); assert np.array_equal(dy.to_host(), yy)
    xi, rr = w.copy(), r.copy()
    O.orc_axpy(n, a, x, xi); O.orc_axpy(n, -a, y, rr); O.orc_pmul(n, rr, dinv, z)
    dy.upload(y); dw.upload(w); dr.upload(r)
    check(lib.liship_cg_update_jacobi_f64(n, a, dx.ptr, dy.ptr, dd.ptr, dw.ptr, dr.ptr, res.ptr, work.ptr, None))
    out = res.to_host()
    assert np.array_equal(dw.to_host(), xi) and np.array_equal(dr.to_host(), rr)
    assert abs(out[0] - np.dot(rr, rr)) <= 1e-14 * np.dot(rr, rr) * max(1.0, np.log2(n + 1))
    assert abs(out[1] - np.dot(rr, z)) <= 1e-14 * np.abs(rr * z).sum() * max(1.0, np.log2(n + 1))


@pytest.mark.parametrize("name", list(CSR_CASES))
@pytest.mark.parametrize("want_sumsq", [0, 1])
def test_spmv_csr_fused_dot(lib, name, want_sumsq):
    """y of the fused form is bit-identical to the plain product; the fused sums agree with the separate
    reduction kernels to 1e-13 relative (different but fixed tree order) and repeat bit for bit."""
    ptr, idx, val = CSR_CASES[name]()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1 if len(idx) else 1)
    rng = np.random.default_rng(11)
    x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    dy = DA.from_host(np.full(n, np.nan), np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    res = DA.from_host(np.full(2, np.nan), np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    got = []
    for _ in range(2):
        check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, want_sumsq,
                                          res.ptr, work.ptr, None))
        got.append(res.to_host())
        assert np.array_equal(dy.to_host(), yref)
    check(lib.liship_csr_plan_destroy(plan))
    assert np.array_equal(got[0][:1 + want_sumsq], got[1][:1 + want_sumsq])
    scale = np.abs(w * yref).sum() + 1e-300
    assert abs(got[0][0] - np.dot(w, yref)) <= 1e-13 * scale
    if want_sumsq:
        assert abs(got[0][1] - np.dot(yref, yref)) <= 1e-13 * np.dot(yref, yref) + 1e-300


@pytest.mark.parametrize("name", ["p3d_64", "rand_5000", "rand_long_rows", "rand_wide_77", "mostly_empty"])
@pytest.mark.parametrize("want_sumsq", [0, 1])
@pytest.mark.parametrize("cuts", [(0.0, 1.0), (0.1, 0.9), (0.0, 0.6), (0.37, 1.0), (0.5, 0.5)])
def test_spmv_csr_fused_dot_in_parts(lib, name, want_sumsq, cuts):
    """interior rows, then the two boundary ranges, one fold: y bit-identical to the plain product, the sums as the
    one-launch fused form to 1e-13 (another grouping of the same terms), repeatable bit for bit"""
    ptr, idx, val = CSR_CASES[name]()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1 if len(idx) else 1)
    rng = np.random.default_rng(12)
    x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    res = DA.from_host(np.full(2, np.nan), np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    lo, hi = int(cuts[0] * n), int(cuts[1] * n)
    got = []
    for _ in range(2):
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        total, used = 0, C.c_int()
        for a, b in ((lo, hi), (0, lo), (hi, n)):
            check(lib.liship_spmv_csr_rows_dot_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, want_sumsq,
                                                   work.ptr, total, C.byref(used), None))
            total += used.value
        check(lib.liship_spmv_csr_dot_finish_f64(total, want_sumsq, res.ptr, work.ptr, None))
        got.append(res.to_host())
        assert np.array_equal(dy.to_host(), yref)
    check(lib.liship_csr_plan_destroy(plan))
    assert np.array_equal(got[0][:1 + want_sumsq], got[1][:1 + want_sumsq])
    scale = np.abs(w * yref).sum() + 1e-300
    assert abs(got[0][0] - np.dot(w, yref)) <= 1e-13 * scale
    if want_sumsq:
        assert abs(got[0][1] - np.dot(yref, yref)) <= 1e-13 * np.dot(yref, yref) + 1e-300


@pytest.mark.parametrize("n", [1, 2, 63, 1000, 4097, (1 << 20) + 3])
def test_gmres_device_chained_kernels(lib, n):
    """The GMRES building blocks that keep their scalars in HBM give the bits of the reference's call sequence
    (lis_vector_axpy / scale chains, lis_solver_gmres.c:219-232, :290-296, :323-329)."""
    rng = np.random.default_rng(n + 11)
    O = orc.lib()
    w, vp, vn = (rng.uniform(-1, 1, n) for _ in range(3))
    h = 0.43219876
    work = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64)
    dh, res = DA.from_host(np.array([h, 0.0])), DA.zeros(2, np.float64)
    dw, dvp, dvn = DA.from_host(w), DA.from_host(vp), DA.from_host(vn)
    # one Gram-Schmidt step: w += (-h) vp ; <w, vn>
    ww = w.copy(); O.orc_axpy(n, -h, vp, ww)
    check(lib.liship_mgs_step_f64(n, dh.ptr, dvp.ptr, dw.ptr, dvn.ptr, res.ptr, work.ptr, None))
    assert np.array_equal(dw.to_host(), ww)
    assert abs(res.to_host()[0] - np.dot(ww, vn)) <= 1e-14 * np.abs(ww * vn).sum() * max(1.0, np.log2(n + 1))
    # the last step: w += (-h) vp ; sum w^2, then w *= 1/sqrt(sum) with the sum read from HBM
    O.orc_axpy(n, -h, vp, ww)
    check(lib.liship_mgs_step_f64(n, dh.ptr, dvp.ptr, dw.ptr, None, res.ptr, work.ptr, None))
    ss = res.to_host()[0]
    assert np.array_equal(dw.to_host(), ww) and abs(ss - np.dot(ww, ww)) <= 1e-14 * np.dot(ww, ww) * max(1.0, np.log2(n + 1))
    check(lib.liship_scale_inv_norm_f64(n, res.ptr, dw.ptr, None))
    O.orc_scale(n, 1.0 / np.sqrt(ss), ww)
    assert np.array_equal(dw.to_host(), ww)
    # linear combinations in the reference's order
    m = 5
    V = [rng.uniform(-1, 1, n) for _ in range(m)]
    coef = rng.uniform(-2, 2, m)
    dV = [DA.from_host(v) for v in V]
    ptrs = (C.c_void_p * m)(*[d.ptr for d in dV])
    dz = DA.from_host(np.full(n, np.nan))
    check(lib.liship_lincomb_f64(n, m, ptrs, coef.ctypes.data, 0, dz.ptr, None))
    z = np.zeros(n); O.orc_axpy(n, coef[0], V[0], z); z = coef[0] * V[0]
    for j in range(1, m):
        O.orc_axpy(n, coef[j], V[j], z)
    assert np.array_equal(dz.to_host(), z)
    # accumulate form with the first vector aliasing the destination: v0 += c0*v0 + c1*v1 + ...
    ptrs2 = (C.c_void_p * m)(*([dV[0].ptr] + [d.ptr for d in dV[1:]]))
    check(lib.liship_lincomb_f64(n, m, ptrs2, coef.ctypes.data, 1, dV[0].ptr, None))
    z = V[0].copy(); O.orc_axpy(n, coef[0], V[0].copy(), z)
    for j in range(1, m):
        O.orc_axpy(n, coef[j], V[j], z)
    assert np.array_equal(dV[0].to_host(), z)


def _hub_columns():
    """6000 rows of 12 random columns, every row also reading column 7 and every second one column 4001: two columns of 6000 / 3000 entries (the lane-per-column
    heap sort beyond the wavefront's LDS stage), the rest short"""
    ptr, idx, val = orc.random_csr(6000, 12, seed=21, ncols=6000, empty_rows=False)
    rows = np.repeat(np.arange(6000), np.diff(ptr))
    extra_r = np.concatenate([np.arange(6000), np.arange(0, 6000, 2)])
    extra_c = np.concatenate([np.full(6000, 7), np.full(3000, 4001)])
    r2, c2 = np.concatenate([rows, extra_r]), np.concatenate([idx, extra_c])
    v2 = np.concatenate([val, np.random.default_rng(22).uniform(-1, 1, len(extra_r))])
    o = np.argsort(r2, kind="stable")
    p2 = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=6000))]).astype(np.int32)
    return p2, c2[o].astype(np.int32), v2[o]


@pytest.mark.parametrize("name", ["p3d_16", "rand_5000", "rand_long_rows", "mostly_empty", "single_row", "rand_wide_77", "fem3_12", "hub_columns"])
def test_csr_transpose_in_scatter_order(lib, name):
    """A^T built in HBM lists every transposed row's entries by their position in the source arrays (the order of
    lis_matvech_csr's scatter): arrays equal to a stable host transposition, whatever the atomics did -- short columns (a lane each), the 33 .. 2048-entry
    columns of finite-element matrices (a wavefront each, ranks counted in LDS: round 6) and hub columns beyond that"""
    ptr, idx, val = _fem(12) if name == "fem3_12" else _hub_columns() if name == "hub_columns" else CSR_CASES[name]()
    n = len(ptr) - 1
    ncols = max(n, int(idx.max()) + 1 if len(idx) else 1)
    nnz = len(idx)
    order = np.argsort(idx, kind="stable")                      # positions grouped by column, ascending inside
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
    tptr_ref = np.zeros(ncols + 1, np.int32)
    np.cumsum(np.bincount(idx, minlength=ncols), out=tptr_ref[1:])
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    tptr, tidx, tval = DA(ncols + 1, np.int32), DA(max(nnz, 1), np.int32), DA(max(nnz, 1), np.float64)
    work = DA(ncols + nnz + 4, np.int32)
    for _ in range(2):
        check(lib.liship_csr_transpose_f64(n, ncols, nnz, dptr.ptr, didx.ptr, dval.ptr, tptr.ptr, tidx.ptr, tval.ptr, work.ptr, None))
        assert np.array_equal(tptr.to_host(), tptr_ref)
        assert np.array_equal(tidx.to_host(nnz), rows[order]) and np.array_equal(tval.to_host(nnz), val[order])


ELL_CODED_CASES = {
    "p3d_20x17x14": (lambda: orc.poisson3d(20, 17, 14, sort_cols=True), True),
    "p3d_8": (lambda: orc.poisson3d(8, 8, 8), True),
    "p3d_odd_33x5x3": (lambda: orc.poisson3d(33, 5, 3), False),                 # odd n: not coded
    "band": (lambda: banded(6000, [-900, -2, -1, 0, 1, 2, 40, 900], 8), True),
    "random": (lambda: orc.random_csr(3000, 9, seed=6, empty_rows=False), False),   # too many diagonals
}


@pytest.mark.parametrize("name", list(ELL_CODED_CASES))
def test_spmv_ell_index_codes(lib, name):
    """ELL with one-byte column codes: coded exactly when it can be, and then the plain product and both reduction
    epilogues return the bits of the 4 B-index kernels (y AND sums: same lanes, same rows, same order)"""
    make, want = ELL_CODED_CASES[name]
    ptr, idx, val = make()
    n = len(ptr) - 1
    mx, eidx, ev = orc.csr2ell(ptr, idx, val)
    rng = np.random.default_rng(41)
    x, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    yref = orc.spmv_ell(n, mx, eidx, ev, x)
    di, dv, dx, dw = DA.from_host(eidx, np.int32), DA.from_host(ev), DA.from_host(x), DA.from_host(w)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    codes, dic, nd = C.c_void_p(), C.c_void_p(), C.c_int()
    check(lib.liship_ell_encode_indices(n, mx, di.ptr, C.byref(codes), C.byref(dic), C.byref(nd), None))
    assert bool(codes.value) == want, (nd.value, want)
    if not codes.value:
        return
    offsets = set((eidx.reshape(mx, n) - np.arange(n)[None, :]).ravel().tolist())
    assert nd.value == len(offsets)
    dy = DA.from_host(np.full(n, np.nan))
    check(lib.liship_spmv_ell_coded_f64(n, mx, codes, dic, dv.ptr, dx.ptr, dy.ptr, None, -1, None, None, None))
    assert np.array_equal(dy.to_host(), yref)
    for sq in (0, 1):
        r1, r2 = DA.from_host(np.full(2, np.nan)), DA.from_host(np.full(2, np.nan))
        dy = DA.from_host(np.full(n, np.nan))
        check(lib.liship_spmv_ell_coded_f64(n, mx, codes, dic, dv.ptr, dx.ptr, dy.ptr, dw.ptr, sq, r1.ptr, work.ptr, None))
        assert np.array_equal(dy.to_host(), yref)
        check(lib.liship_spmv_ell_dot_f64(n, mx, di.ptr, dv.ptr, dx.ptr, dy.ptr, dw.ptr, sq, r2.ptr, work.ptr, None))
        assert np.array_equal(r1.to_host()[:1 + sq], r2.to_host()[:1 + sq])
    check(lib.liship_free(codes))
    check(lib.liship_free(dic))


@pytest.mark.parametrize("fmt", ["ell", "dia"])
@pytest.mark.parametrize("want_sumsq", [0, 1])
@pytest.mark.parametrize("grid", [(20, 17, 14), (8, 8, 8), (33, 5, 2)])
def test_ell_dia_fused_dot(lib, fmt, want_sumsq, grid):
    """ELL / DIA products with the reduction epilogue: y bit-identical to the plain product, sums as the separate
    reductions (1e-13 relative), repeatable bit for bit"""
    ptr, idx, val = orc.poisson3d(*grid, sort_cols=True)
    n = len(ptr) - 1
    rng = np.random.default_rng(5)
    x, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    dx, dw, dy = DA.from_host(x), DA.from_host(w), DA.from_host(np.full(n, np.nan))
    work, res = DA(lib.liship_reduce_work_bytes() // 8, np.float64), DA.from_host(np.full(2, np.nan))
    if fmt == "ell":
        mx, eidx, ev = orc.csr2ell(ptr, idx, val)
        di, dv = DA.from_host(eidx, np.int32), DA.from_host(ev)
        yref = orc.spmv_ell(n, mx, eidx, ev, x)
        call = lambda: lib.liship_spmv_ell_dot_f64(n, mx, di.ptr, dv.ptr, dx.ptr, dy.ptr, dw.ptr, want_sumsq, res.ptr, work.ptr, None)
    else:
        nnd, off, dval = orc.csr2dia(ptr, idx, val)
        di, dv = DA.from_host(off, np.int32), DA.from_host(dval)
        yref = orc.spmv_dia(n, nnd, off, dval, x)
        call = lambda: lib.liship_spmv_dia_dot_f64(n, n, nnd, di.ptr, dv.ptr, dx.ptr, dy.ptr, dw.ptr, want_sumsq, res.ptr, work.ptr, None)
    rc = call()
    if n % 2:
        assert rc == -1                                        # odd n: the caller falls back to product + dot
        return
    check(rc)
    first = res.to_host()
    assert np.array_equal(dy.to_host(), yref)
    check(call())
    assert np.array_equal(res.to_host()[:1 + want_sumsq], first[:1 + want_sumsq])
    assert abs(first[0] - np.dot(w, yref)) <= 1e-13 * np.abs(w * yref).sum()
    if want_sumsq:
        assert abs(first[1] - np.dot(yref, yref)) <= 1e-13 * np.dot(yref, yref)


def test_full_size_512_properties(lib):
    """BASELINE's full size (512^3 rows, 938 M non-zeros) through size-independent properties, no CPU pass:
    A*1 equals the closed form everywhere (exact: small integers), the row-range launches and the fused-dot form
    write the bits of the plain launch, <A x, y> == <x, A y> to rounding (A is symmetric), linearity in x."""
    N = 512
    n = N ** 3
    nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
    assert nnz == 7 * n - 6 * N * N == 937951232
    dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
    check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    x, y, y2, b = (DA(n, np.float64) for _ in range(4))

    def spmv(src, dst):
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, src.ptr, dst.ptr, None))

    def nrm1(v):
        check(lib.liship_nrm1_f64(n, v.ptr, res.ptr, work.ptr, None))
        return res.to_host()[0]

    def dot(u, v):
        check(lib.liship_dot_f64(n, u.ptr, v.ptr, res.ptr, work.ptr, None))
        return res.to_host()[0]

    # A*1: closed form, exact
    check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
    spmv(x, y)
    check(lib.liship_sumsq_f64(n, y.ptr, res.ptr, work.ptr, None))
    assert res.to_host()[0] == 6.0 * (N - 2) ** 2 + 48.0 * (N - 2) + 72.0
    check(lib.liship_poisson3d_rhs(N, N, N, 0, n, b.ptr, None))
    check(lib.liship_axpy_f64(n, -1.0, b.ptr, y.ptr, None))
    assert nrm1(y) == 0.0
    # a non-trivial x: x_i = frac(i * golden ratio) - 0.5, built in pieces to keep the host footprint small
    chunk = 1 << 24
    for s in range(0, n, chunk):
        part = np.modf(np.arange(s, min(n, s + chunk), dtype=np.float64) * 0.6180339887498949)[0] - 0.5
        check(lib.liship_memcpy_h2d(x.ptr + 8 * s, part.ctypes.data, part.nbytes, None))
        check(lib.liship_device_synchronize())
    spmv(x, y)
    # row-range launches (the multi-GPU overlap path) write the same bits
    check(lib.liship_memset(y2.ptr, 0xff, 8 * n, None))
    for lo, hi in ((0, 262144), (262144, n - 300000), (n - 300000, n)):
        check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y2.ptr, None))
    check(lib.liship_axpy_f64(n, -1.0, y.ptr, y2.ptr, None))
    assert nrm1(y2) == 0.0
    # the fused-dot form: same y, and its <x,y> agrees with the separate reduction
    check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y2.ptr, x.ptr, 1, res.ptr, work.ptr, None))
    fused = res.to_host().copy()
    xAx = dot(x, y)
    assert abs(fused[0] - xAx) <= 1e-12 * abs(xAx)
    check(lib.liship_sumsq_f64(n, y.ptr, res.ptr, work.ptr, None))
    assert abs(fused[1] - res.to_host()[0]) <= 1e-12 * fused[1]
    check(lib.liship_axpy_f64(n, -1.0, y.ptr, y2.ptr, None))
    assert nrm1(y2) == 0.0
    # symmetry: <A x, b> == <x, A b>  (b = A*1 as the second vector), and linearity: A(2x) == 2 A x exactly
    spmv(b, y2)
    lhs, rhs = dot(y, b), dot(x, y2)
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), 1.0)
    check(lib.liship_scale_f64(n, 2.0, x.ptr, None))
    spmv(x, y2)
    check(lib.liship_axpy_f64(n, -2.0, y.ptr, y2.ptr, None))
    assert nrm1(y2) == 0.0
    check(lib.liship_csr_plan_destroy(plan))


def test_team_kernels_at_scale(lib):
    """the two team-of-lanes kernels of round 3 at the sizes their timings are quoted on, against the oracle bit for bit: the 27-point stencil
    with random coefficients at 160^3 (4.1 M rows, 109 M non-zeros: spmv_csr_pattern_team_kernel vs the one-lane-per-row pattern kernel vs
    lis_matvec_csr's order) and the 3-dofs-per-node finite-element pattern in 3x3 BSR at 64^3 nodes (0.79 M rows, 61 M non-zeros:
    spmv_bsr_team_kernel vs the two-phase tile kernel vs lis_matvec_bsr's order)"""
    ptr, idx, val = stencil_box_variable_coefficients((160, 160, 160), 12)
    n = len(ptr) - 1
    x = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_row_patterns(plan) == 27 and lib.liship_csr_plan_team_records(plan) == 1 and lib.liship_csr_plan_value_records(plan) == 0
    for variant in (0, 0x2000):
        lib.liship_spmv_csr_set_variant(variant)
        dy = DA.from_host(np.full(n, np.nan), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), hex(variant)
    lib.liship_spmv_csr_set_variant(0)
    check(lib.liship_csr_plan_destroy(plan))
    del dptr, didx, dval, dx, dy

    ptr, idx, val = orc.fem3(64, 3)[:3]
    n = len(ptr) - 1
    nr, bptr, bidx, bv = orc.csr2bsr(ptr, idx, val, 3, 3)
    assert len(bidx) / nr > 16                      # long block rows: the dispatcher takes the team kernel
    xx = np.zeros(nr * 3 + 3)
    xx[:n] = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    ref = orc.spmv_bsr(n, nr, 3, 3, bptr, bidx, bv, xx)
    a, b, c, dx = DA.from_host(bptr), DA.from_host(bidx), DA.from_host(bv), DA.from_host(xx)
    for team in (1, 0):
        lib.liship_spmv_bsr_set_team(team)
        dy = DA.from_host(np.full(nr * 3, np.nan))
        check(lib.liship_spmv_bsr_nnz_f64(nr, len(bidx), 3, 3, a.ptr, b.ptr, c.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host(n).view(np.uint64), ref.view(np.uint64)), team
    lib.liship_spmv_bsr_set_team(1)


def test_full_size_512_nine_forms_bit_equal(lib):
    """BASELINE's full size with a NON-TRIVIAL x through every form of the CSR product the plan can choose -- 4 B indices,
    one-byte column codes, row patterns (general kernel and the 32 B-record kernel), value records one and two rows per lane and with
    the dominant pattern's gathers speculated (two and four rows per lane):
    nrm1(y_form - y_4B) == 0 at 512^3, and the fused <x, A x>, ||A x||^2 of every form are the same bits.  The 4 B form is the
    contract kernel (lis_matvec_csr.c:97-109 restated), itself bit-compared with the oracle at the sizes the oracle finishes."""
    N = 512
    n = N ** 3
    nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
    dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
    check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_coded(plan) == 7 and lib.liship_csr_plan_row_patterns(plan) == 27
    assert lib.liship_csr_plan_pattern_records(plan) == 1 and lib.liship_csr_plan_value_records(plan) == 1
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    x, y0, y = (DA(n, np.float64) for _ in range(3))
    chunk = 1 << 24
    for s in range(0, n, chunk):          # x_i = frac(i * golden ratio) - 0.5: no two neighbours alike, both signs
        part = np.modf(np.arange(s, min(n, s + chunk), dtype=np.float64) * 0.6180339887498949)[0] - 0.5
        check(lib.liship_memcpy_h2d(x.ptr + 8 * s, part.ctypes.data, part.nbytes, None))
        check(lib.liship_device_synchronize())

    def select(on):
        lib.liship_spmv_csr_set_index_codes(1 if on else 0)
        lib.liship_spmv_csr_set_row_patterns(1 if on >= 2 else 0)
        lib.liship_spmv_csr_set_row_values(1 if on >= 4 else 0)
        lib.liship_spmv_csr_set_variant(VARIANT_OF_FORM.get(on, 0))

    dots = {}
    try:
        for on in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            select(on)
            dst = y0 if on == 0 else y
            check(lib.liship_memset(dst.ptr, 0xff, 8 * n, None))
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, dst.ptr, None))
            if on:
                check(lib.liship_axpy_f64(n, -1.0, y0.ptr, y.ptr, None))
                check(lib.liship_nrm1_f64(n, y.ptr, res.ptr, work.ptr, None))
                assert res.to_host()[0] == 0.0, on
            check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
            check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 1, res.ptr, work.ptr, None))
            dots[on] = res.to_host().copy()
            check(lib.liship_axpy_f64(n, -1.0, y0.ptr, y.ptr, None))
            check(lib.liship_nrm1_f64(n, y.ptr, res.ptr, work.ptr, None))
            assert res.to_host()[0] == 0.0, ("fused", on)
        # the ninth form (round 5): the contract kernel again with the XCD strips OFF -- a permutation of its row blocks, not of any bit (y, and the fused dots,
        # whose partials stay with their blocks)
        select(0)
        check(lib.liship_spmv_csr_set_xcd_strips(0))
        check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))
        check(lib.liship_axpy_f64(n, -1.0, y0.ptr, y.ptr, None))
        check(lib.liship_nrm1_f64(n, y.ptr, res.ptr, work.ptr, None))
        assert res.to_host()[0] == 0.0, "strips off"
        check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
        check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 1, res.ptr, work.ptr, None))
        assert np.array_equal(res.to_host(), dots[0]), ("strips off", res.to_host(), dots[0])
    finally:
        select(4)
        lib.liship_spmv_csr_set_variant(0)
        check(lib.liship_spmv_csr_set_xcd_strips(1))
    # the product is not trivially zero, and x^T A x > 0 (A is positive definite)
    check(lib.liship_nrm1_f64(n, y0.ptr, res.ptr, work.ptr, None))
    assert res.to_host()[0] > 1e6 and dots[0][0] > 0.0
    # ... and the chain to the ORACLE is closed at this size too: three slabs of 2^20 rows -- the first rows (a boundary face), rows across
    # plane boundaries in the middle, the last rows -- are generated by the oracle's own generator (test/test3.c:114-127 restated), multiplied
    # by the oracle (lis_matvec_csr.c:97-109 restated) and compared with the same rows of y0 IN EVERY BIT; every other form equals y0 above.
    # The device-made matrix is compared with the generator's arrays on the same rows on the way.
    slab, mn = 1 << 20, N * N
    hp = np.empty(slab + 1, np.int32)
    for r0 in (0, (n // 2) - (slab // 2) + 12345, n - slab):
        r1 = r0 + slab
        ptr_s, idx_s, val_s = orc.poisson3d(N, N, N, is_=r0, ie=r1)
        check(lib.liship_memcpy_d2h(hp.ctypes.data, dptr.ptr + 4 * r0, hp.nbytes, None))
        check(lib.liship_device_synchronize())
        k0, k1 = int(hp[0]), int(hp[-1])
        assert np.array_equal(hp - hp[0], ptr_s)
        hidx, hval = np.empty(k1 - k0, np.int32), np.empty(k1 - k0, np.float64)
        check(lib.liship_memcpy_d2h(hidx.ctypes.data, didx.ptr + 4 * k0, hidx.nbytes, None))
        check(lib.liship_memcpy_d2h(hval.ctypes.data, dval.ptr + 8 * k0, hval.nbytes, None))
        check(lib.liship_device_synchronize())
        assert np.array_equal(hidx, idx_s) and np.array_equal(hval, val_s)
        lo, hi = max(0, r0 - mn), min(n, r1 + mn)          # the columns these rows reach
        xwin, yslab = np.empty(hi - lo, np.float64), np.empty(slab, np.float64)
        check(lib.liship_memcpy_d2h(xwin.ctypes.data, x.ptr + 8 * lo, xwin.nbytes, None))
        check(lib.liship_memcpy_d2h(yslab.ctypes.data, y0.ptr + 8 * r0, yslab.nbytes, None))
        check(lib.liship_device_synchronize())
        yref = orc.spmv_csr(ptr_s, idx_s - lo, val_s, xwin)
        assert np.array_equal(yslab.view(np.uint64), yref.view(np.uint64)), r0
    # the forms that share a row-block geometry share their partial sums (coded plans rebuild the split at 256 / 2048, the 4 B form keeps
    # the 192 / 1408 one of the bare plan only when no codes exist -- here every form runs on the coded plan's blocks)
    for on in (1, 2, 3, 4, 5, 8):
        assert np.array_equal(dots[on], dots[0]), (on, dots[on], dots[0])
    # the dominant-pattern product's own epilogue (a partial per tile / per chunk of 512 rows, forms 7 / 6): the same sums to rounding
    for on in (6, 7):
        assert np.all(np.abs(dots[on] - dots[0]) <= 1e-13 * np.abs(dots[0])), (on, dots[on], dots[0])
    check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("n", [1, 2, 63, 1000, 1 << 20, (1 << 20) + 1])
def test_uniform_jacobi_passes(lib, n):
    """the CG passes that take 1/diag as one double give the bits of the passes that read an array holding that double everywhere,
    and liship_count_ne_f64 counts the elements that differ from it in any bit (-0.0 against 0.0 included)"""
    rng = np.random.default_rng(n)
    dc = 1.0 / 6.0
    r, p, x, q = (rng.uniform(-1, 1, n) for _ in range(4))
    scal = DA.from_host(np.array([0.37, -0.81, 0.59]), np.float64)          # alpha, beta, -alpha
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    dinv = DA.from_host(np.full(n, dc), np.float64)
    res = DA.from_host(np.full(2, np.nan), np.float64)
    check(lib.liship_count_ne_f64(n, dinv.ptr, dc, res.ptr, work.ptr, None))
    assert res.to_host()[0] == 0.0
    bumped = np.full(n, dc)
    bumped[n // 2] = np.nextafter(dc, 1.0)
    if n > 2:
        bumped[0] = -dc
    dbump, dzero = DA.from_host(bumped, np.float64), DA.from_host(np.full(n, -0.0), np.float64)
    check(lib.liship_count_ne_f64(n, dbump.ptr, dc, res.ptr, work.ptr, None))
    assert res.to_host()[0] == (2.0 if n > 2 else 1.0)
    check(lib.liship_count_ne_f64(n, dzero.ptr, 0.0, res.ptr, work.ptr, None))
    assert res.to_host()[0] == float(n)
    outs = []
    for uniform in (0, 1):
        dr, dp, dx, dq = (DA.from_host(v, np.float64) for v in (r, p, x, q))
        ss = scal.ptr
        for first in (1, 0):                     # the first iteration's form (no x update), then the general one
            pa = None if first else ss
            if uniform:
                check(lib.liship_cg_direction_uniform_dev_f64(n, pa, ss + 8, dr.ptr, dc, dp.ptr, dx.ptr, None))
            else:
                check(lib.liship_cg_direction_dev_f64(n, pa, ss + 8, dr.ptr, dinv.ptr, dp.ptr, dx.ptr, None))
        res = DA.from_host(np.full(2, np.nan), np.float64)
        if uniform:
            check(lib.liship_cg_residual_jacobi_uniform_dev_f64(n, ss + 16, dq.ptr, dc, dr.ptr, res.ptr, work.ptr, None))
        else:
            check(lib.liship_cg_residual_jacobi_dev_f64(n, ss + 16, dq.ptr, dinv.ptr, dr.ptr, res.ptr, work.ptr, None))
        outs.append((dp.to_host(), dx.to_host(), dr.to_host(), res.to_host()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    assert not np.array_equal(outs[0][1], x) and not np.array_equal(outs[0][2], r)       # the passes did run


def test_reduction_order_is_pinned(lib):
    """tests/golden/reduction_bits.json (made on an MI355X by this library: make_golden_reduction_bits.py) holds the bits of dot,
    sum of squares, the two-result dot and the reduction epilogue of the CSR product on seeded inputs.  The order of these
    reductions -- wavefront butterfly, wavefronts of a workgroup in order, tree fold over the partials -- is part of what the
    Krylov iteration counts hang on: a kernel change that is meant to keep it must reproduce every bit."""
    import importlib.util
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_reduction_bits", os.path.join(here, "golden", "make_golden_reduction_bits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "reduction_bits.json")))
    got = mod.measure(lib)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


def _structured_random(seed):
    """a random matrix of the structured kind the plan's coders look for: rows drawn from a few (offset subset, value set)
    templates, with knobs for everything that switches a coder on or off -- the number of diagonals and templates, empty rows,
    rows longer than 7, values from a palette / per template / fully random, columns beyond n"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 1000, 2500, 4099, 9000]))
    nd = int(rng.integers(1, 10)) if rng.random() < 0.7 else int(rng.integers(10, 30))      # up to 7: the 32 B records; 8..32: the wide ones
    reach = min(n, 300) if rng.random() < 0.7 else min(n, 12)                               # a narrow band keeps the boundary patterns few
    offs = np.unique(np.concatenate([[0], rng.integers(-reach, reach + 1, nd - 1)])) if nd > 1 else np.array([0])
    ncols = n + (int(rng.integers(1, 50)) if rng.random() < 0.3 else 0)
    ntemp = int(rng.integers(1, 7))
    temps = []
    for _ in range(ntemp):
        keep = rng.random(len(offs)) < rng.choice([0.5, 0.8, 1.0])
        if rng.random() < 0.15:
            keep[:] = False                                 # an empty-row template
        mode = rng.choice(["palette", "template", "random"])
        vals = rng.choice([-1.0, 6.0, 0.0, 0.25, -0.0], len(offs)) if mode == "palette" else rng.uniform(-2, 2, len(offs))
        temps.append((keep, vals, mode))
    which = rng.integers(0, ntemp, n) if rng.random() < 0.5 else np.minimum(np.arange(n) * ntemp // max(n, 1), ntemp - 1)
    ptr, idx, val = [0], [], []
    for r in range(n):
        keep, vals, mode = temps[which[r]]
        for k, o in enumerate(offs):
            c = r + int(o)
            if keep[k] and 0 <= c < ncols:
                idx.append(c)
                val.append(rng.uniform(-2, 2) if mode == "random" else vals[k])
        ptr.append(len(idx))
    return np.array(ptr, np.int32), np.array(idx, np.int32), np.array(val, np.float64), ncols


def _stencil_random(seed):
    """a random stencil on a random small grid -- a subset of the 27 box offsets (so that runs of 1, 2 and 3 neighbouring columns occur) plus, sometimes, far
    offsets; Dirichlet truncation; values constant per offset, per row, or constant with a few rows of their own; a few rows moved off the pattern"""
    rng = np.random.default_rng(seed)
    dims = tuple(int(v) for v in rng.choice([1, 3, 5, 8, 13, 21, 34, 70], 3))
    if dims[0] * dims[1] * dims[2] < 600:
        dims = (dims[0] + 7, dims[1] + 9, dims[2] + 17)
    pts = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) == (0, 0, 0) or rng.random() < rng.choice([0.45, 0.8, 1.0])]
    if rng.random() < 0.3:
        pts += [(0, 0, int(k)) for k in rng.choice([-4, -3, 3, 4, 6], 2, replace=False)]
    ptr, idx, val = stencil_points(dims, sorted(set(pts)), seed)
    n = len(ptr) - 1
    mode = rng.choice(["constant", "random", "mostly_constant"])
    if mode != "random":
        offs = idx - np.repeat(np.arange(n), np.diff(ptr))
        palette = {int(o): float(v) for o, v in zip(np.unique(offs), rng.choice([-1.0, 26.0, 0.5, -0.0, 0.0, 3.25], len(np.unique(offs))))}
        cval = np.array([palette[int(o)] for o in offs])
        if mode == "mostly_constant":
            rows = rng.choice(n, max(1, n // 50), replace=False)
            for r in rows:
                cval[ptr[r]:ptr[r + 1]] = rng.uniform(-2, 2, ptr[r + 1] - ptr[r])
        val = cval
    if rng.random() < 0.4:                                  # rows off the pattern: the last entry moves
        for r in rng.choice(n, max(1, n // 40), replace=False):
            if ptr[r + 1] > ptr[r]:
                k = ptr[r + 1] - 1
                idx[k] = min(n - 1, idx[k] + int(rng.integers(2, 9)))
                order = np.argsort(idx[ptr[r]:ptr[r + 1]], kind="stable")
                idx[ptr[r]:ptr[r + 1]] = idx[ptr[r]:ptr[r + 1]][order]
                val[ptr[r]:ptr[r + 1]] = val[ptr[r]:ptr[r + 1]][order]
    return ptr, idx.astype(np.int32), np.asarray(val, np.float64)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LIS_AMD_FUZZ_SEEDS", "60"))))
def test_team_and_staged_kernels_on_random_stencils(lib, seed):
    """random stencils on random grids through whatever the plan builds for them -- team records, runs and slots, wide records with a dominant pattern, masks,
    foreign rows -- in every form of the product (x staged / gathered / a lane per row / round 2's kernels), whole and in row ranges: the oracle's bits"""
    ptr, idx, val = _stencil_random(seed)
    n = len(ptr) - 1
    x = np.random.default_rng(2000 + seed).uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    state = (lib.liship_csr_plan_row_patterns(plan), lib.liship_csr_plan_team_form(plan), lib.liship_csr_plan_value_records(plan), lib.liship_csr_plan_wide_dominant(plan))
    try:
        for values_on in (1, 0):
            lib.liship_spmv_csr_set_row_values(values_on)
            for variant in (0, 0x4000, 0x2000):
                lib.liship_spmv_csr_set_variant(variant)
                dy = DA.from_host(np.full(n, np.nan), np.float64)
                check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (seed, state, values_on, hex(variant))
                a, b = n // 3 + 1, n - n // 5 - 3
                dy = DA.from_host(np.full(n, np.nan), np.float64)
                for lo, hi in ((a, b), (0, a), (b, n)):
                    check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (seed, state, "rows", values_on, hex(variant))
    finally:
        lib.liship_spmv_csr_set_row_values(1)
        lib.liship_spmv_csr_set_variant(0)
        check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LIS_AMD_FUZZ_SEEDS", "60"))))      # more seeds: LIS_AMD_FUZZ_SEEDS=1000
def test_plan_coders_on_random_structured_matrices(lib, seed):
    """whatever the plan decides to keep -- codes, row patterns, 32 B records, value records (refined or not) -- every form of the
    product and of the fused dots must give the bits of the 4 B-index kernel, which must give the oracle's"""
    ptr, idx, val, ncols = _structured_random(seed)
    n = len(ptr) - 1
    rng = np.random.default_rng(1000 + seed)
    x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr = DA.from_host(ptr, np.int32)
    didx = DA.from_host(idx if len(idx) else np.zeros(1, np.int32), np.int32)
    dval = DA.from_host(val if len(val) else np.zeros(1), np.float64)
    dx, dw = DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    state = (lib.liship_csr_plan_coded(plan), lib.liship_csr_plan_row_patterns(plan), lib.liship_csr_plan_pattern_records(plan),
             lib.liship_csr_plan_value_records(plan))
    assert (state[3] == 1) <= state[2] <= (1 if state[1] else 0) <= (1 if state[0] else 0) and (state[3] != 2 or (state[1] and not state[2])), state
    dots, wide_dots = [], []
    wide = lib.liship_csr_plan_wide_dominant(plan) == 1      # the staged wide-record kernel: an epilogue of its own (one partial per 256 rows)
    dom = lib.liship_csr_plan_dominant_pattern(plan) == 1    # ... and the dominant-pattern product of a plan with value records (one per tile / 512 rows; 0x4000: the row blocks' sums)
    try:
        for codes, pats, vals_on, variant in ((0, 0, 0, 0), (1, 0, 0, 0), (1, 1, 0, 0x2000), (1, 1, 0, 0), (1, 1, 1, 0x20000000), (1, 1, 1, 0x20004000), (1, 1, 1, 0x10000000), (1, 1, 1, 0x4000), (1, 1, 1, 0)):
            lib.liship_spmv_csr_set_index_codes(codes)
            lib.liship_spmv_csr_set_row_patterns(pats)
            lib.liship_spmv_csr_set_row_values(vals_on)
            lib.liship_spmv_csr_set_variant(variant)
            dy = DA.from_host(np.full(max(n, 1), np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host()[:n], yref), (state, codes, pats, vals_on, variant)
            lo, hi = n // 3, n - n // 4
            dy = DA.from_host(np.full(max(n, 1), np.nan), np.float64)
            for a, b in ((lo, hi), (0, lo), (hi, n)):
                check(lib.liship_spmv_csr_rows_f64(plan, a, b, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host()[:n], yref), (state, "rows", codes, pats, vals_on, variant)
            res = DA.from_host(np.full(2, np.nan), np.float64)
            dy = DA.from_host(np.full(max(n, 1), np.nan), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1, res.ptr, work.ptr, None)
            if rc == 0:
                assert np.array_equal(dy.to_host()[:n], yref)
                own = (wide and (codes, pats, vals_on, variant) == (1, 1, 1, 0)) or (dom and (codes, pats, vals_on) == (1, 1, 1) and variant in (0, 0x10000000))
                (wide_dots if own else dots).append(res.to_host().copy())
    finally:
        lib.liship_spmv_csr_set_index_codes(1)
        lib.liship_spmv_csr_set_row_patterns(1)
        lib.liship_spmv_csr_set_row_values(1)
        lib.liship_spmv_csr_set_variant(0)
        check(lib.liship_csr_plan_destroy(plan))
    for d in dots[1:]:
        assert np.array_equal(d, dots[0]), state
    for d in wide_dots:                                     # the same sums to rounding
        scale = float(np.abs(w).sum() * (np.abs(yref).max() if n else 0.0) + np.dot(yref, yref)) + 1e-300
        assert dots and np.all(np.abs(d - dots[0]) <= 1e-12 * scale), (state, d, dots[0])


@pytest.mark.parametrize("dims", [(128, 128, 128), (40, 256, 128), (72, 96, 200)])
def test_xcd_strips_permute_blocks_not_bits(lib, dims):
    """Round 4: the 7-offset pattern kernel (values streamed) walks its row blocks in XCD strips on structured grids -- every XCD one eighth of every plane, plane after
    plane (xcd_strip_unit).  A permutation of the blocks: y must be the oracle's bits with the strips on and off, whole and in the row ranges of a multi-rank slab
    (interior / boundary parts), and the fused dots -- a partial per row block, folded in block order -- the same bits either way.  Grids whose planes are 64+ row
    blocks (strips active: 128 x 128 and 256 x 128 planes) and one whose planes are too small (96 x 200: natural order)."""
    l, m, n_ = dims
    ptr, idx, val = orc.poisson3d(l, m, n_)
    n = len(ptr) - 1
    rng = np.random.default_rng(l * 7 + m)
    val = val * rng.uniform(0.5, 1.5, len(val))                 # varying coefficients: no value records, the values are streamed
    x = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_pattern_records(plan) == 1 and lib.liship_csr_plan_value_records(plan) == 0
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    mn = m * n_
    dots = {}
    try:
        for strips in (1, 0):
            check(lib.liship_spmv_csr_set_xcd_strips(strips))
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            for lo, hi in ((mn, n - mn), (0, mn), (n - mn, n)):          # a slab's interior rows first, then its two boundary planes
                check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips, "ranges")
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dx.ptr, 1, res.ptr, work.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips, "fused")
            dots[strips] = res.to_host().copy()
        assert np.array_equal(dots[0].view(np.uint64), dots[1].view(np.uint64)), dots
        assert abs(dots[1][0] - np.dot(x, yref)) <= 1e-12 * np.abs(x * yref).sum()
    finally:
        check(lib.liship_spmv_csr_set_xcd_strips(1))
        check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("dims", [(128, 128, 128), (40, 256, 128), (72, 96, 200)])
@pytest.mark.parametrize("form", ["bare_4B", "full_4B", "full_codes"])
def test_xcd_strips_of_the_index_streaming_kernels(lib, dims, form):
    """Round 5: the kernels that stream index[] (spmv_csr_rowgather_kernel, the contract's 12 B per non-zero) or the one-byte codes (spmv_csr_coded_kernel) walk their
    row blocks in the same XCD strips.  A plan without row patterns learns the plane from the band of the matrix (liship_csr_plan_scan_band: the largest |column - row|,
    reached by most rows).  y must be the oracle's bits with the strips on and off, whole, in a slab's row ranges and with the fused dots, whose sums must not move."""
    l, m, n_ = dims
    ptr, idx, val = orc.poisson3d(l, m, n_)
    n = len(ptr) - 1
    val = val * np.random.default_rng(l + m).uniform(0.5, 1.5, len(val))
    x = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    yref = orc.spmv_csr(ptr, idx, val, x)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    if form != "bare_4B":
        check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
        check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_scan_band(plan, dptr.ptr, didx.ptr, None))
    mn = m * n_
    assert lib.liship_csr_plan_strip_rows(plan) == mn           # the band of the bare plan = the largest pattern offset of the full one
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    dots = {}
    try:
        lib.liship_spmv_csr_set_index_codes(1 if form == "full_codes" else 0)
        lib.liship_spmv_csr_set_row_patterns(0)
        for strips in (1, 0):
            check(lib.liship_spmv_csr_set_xcd_strips(strips))
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips)
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            for lo, hi in ((mn, n - mn), (0, mn), (n - mn, n)):
                check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips, "ranges")
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dx.ptr, 1, res.ptr, work.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yref.view(np.uint64)), (dims, strips, "fused")
            dots[strips] = res.to_host().copy()
        assert np.array_equal(dots[0].view(np.uint64), dots[1].view(np.uint64)), dots
        assert abs(dots[1][0] - np.dot(x, yref)) <= 1e-12 * np.abs(x * yref).sum()
    finally:
        lib.liship_spmv_csr_set_index_codes(1)
        lib.liship_spmv_csr_set_row_patterns(1)
        check(lib.liship_spmv_csr_set_xcd_strips(1))
        check(lib.liship_csr_plan_destroy(plan))


def test_scan_band_leaves_irregular_matrices_alone(lib):
    """the band is only a plane when most rows reach it: a random matrix (every row its own largest offset) and a band matrix with a few far outliers keep the natural order"""
    for name, (ptr, idx, val) in (("random", orc.random_csr(70000, 9, seed=11, empty_rows=False)), ("p1d", orc.poisson1d(70000))):
        n = len(ptr) - 1
        if name == "p1d":
            idx = idx.copy()
            idx[ptr[5]] = n - 1                                  # one far entry: the largest |column - row|, reached by one row
        dptr, didx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32)
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
        check(lib.liship_csr_plan_scan_band(plan, dptr.ptr, didx.ptr, None))
        assert lib.liship_csr_plan_strip_rows(plan) == 0, name
        check(lib.liship_csr_plan_destroy(plan))


def _box27_per_slot(dims, values):
    """the 27-point box stencil with ONE value per slot (dz, dy, dx) -- constant coefficients, but 27 different ones: a wrong slot order shows in the bits"""
    ptr, idx, val = stencil_box(dims)
    nz, ny, nx = dims
    rows = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    off = idx.astype(np.int64) - rows
    dz = np.rint(off / (ny * nx)).astype(np.int64)
    rem = off - dz * ny * nx
    dy = np.rint(rem / nx).astype(np.int64)
    dx = rem - dy * nx
    return ptr, idx, values[((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)).astype(np.int64)]


@pytest.mark.parametrize("dims,march", [((9, 8, 128), 2), ((10, 12, 128), 2), ((9, 8, 128), 3), ((11, 16, 256), 2), ((8, 8, 256), 2),
                                        # partial tiles (round 5): lines that are not a multiple of 128 long, planes whose lines are not a multiple of a tile's
                                        ((9, 10, 192), 2), ((10, 7, 132), 2), ((9, 13, 200), 2), ((8, 9, 254), 2), ((9, 6, 136), 3), ((9, 17, 128), 2), ((9, 11, 320), 3)])
@pytest.mark.parametrize("values", ["hpcg", "per_slot"])
def test_box27_marching_kernel_bit_exact(lib, dims, march, values):
    """Round 5: the 27-point box stencil with constant coefficients on a grid that is a box -- the matrix of the reference's spmvtest3b (test/spmvtest3b.c:136-160)
    and of HPCG -- walks the planes of 128-column tiles (spmv_csr_box27_march_kernel: each x loaded once, the planes before in registers, neighbours outside the grid as
    signed zeros in the halo).  y must be the oracle's bits -- with 27 DIFFERENT values a wrong slot or order cannot hide --, whole, in ranges of whole planes, with
    the fused dots; tiles of 8 lines (march 2) and of 4 (grids whose planes have 12 lines; march 3 forces them); x with infinities, NaN and signed zeros; and the
    same bits as the staged kernel the plan runs with marching off."""
    nz, ny, nx = dims
    if values == "hpcg":
        ptr, idx, val = stencil_box(dims)
    else:                                                   # 27 different values of BOTH signs (and one explicit zero): a stiffness matrix on a regular mesh
        v27 = np.random.default_rng(3).uniform(-1.5, 1.5, 27)
        v27[13] = 26.5
        v27[7] = 0.0
        ptr, idx, val = _box27_per_slot(dims, v27)
    n = len(ptr) - 1
    x = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    x[::7] = 0.0
    x[3::11] = -0.0
    for pos, v in ((0, np.inf), (n - 1, -np.inf), (n // 2, np.nan), (n // 3, np.inf), (nx * 3 + 5, np.nan), (ny * nx * 2 + 127, -np.inf)):
        x[pos] = v
    yref = orc.spmv_csr(ptr, idx, val, x)
    wv = np.random.default_rng(9).uniform(-1, 1, n)
    dptr, didx, dval, dx, dw = (DA.from_host(a, t) for a, t in ((ptr, np.int32), (idx, np.int32), (val, np.float64), (x, np.float64), (wv, np.float64)))
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_wide_dominant(plan) == 1
    work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    mn = ny * nx
    try:
        check(lib.liship_spmv_csr_set_dom_march(march))
        assert lib.liship_csr_plan_box27(plan) == 1
        dy = DA.from_host(np.full(n, 7.0), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        y = dy.to_host()
        assert np.array_equal(y.view(np.uint64)[~np.isnan(yref)], yref.view(np.uint64)[~np.isnan(yref)]) and np.array_equal(np.isnan(y), np.isnan(yref))
        dy = DA.from_host(np.full(n, 7.0), np.float64)
        for lo, hi in ((mn, n - mn), (0, mn), (n - mn, n)):                # a slab's interior planes, then its boundary planes
            check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        y2 = dy.to_host()
        assert np.array_equal(y2.view(np.uint64)[~np.isnan(yref)], yref.view(np.uint64)[~np.isnan(yref)])
        # fused dots on a finite x (w = x and w a vector of its own): y again the oracle's, the sums to rounding
        xf = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
        yf = orc.spmv_csr(ptr, idx, val, xf)
        check(lib.liship_memcpy_h2d(dx.ptr, xf.ctypes.data, xf.nbytes, None))
        for wname, wd, wh in (("x", dx, xf), ("w", dw, wv)):
            dy = DA.from_host(np.full(n, 7.0), np.float64)
            check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, wd.ptr, 1, res.ptr, work.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), yf.view(np.uint64)), wname
            got = res.to_host()
            assert abs(got[0] - np.dot(wh, yf)) <= 1e-12 * np.abs(wh * yf).sum() and abs(got[1] - np.dot(yf, yf)) <= 1e-12 * np.dot(yf, yf), (wname, got)
        check(lib.liship_spmv_csr_set_dom_march(0))
        assert lib.liship_csr_plan_box27(plan) == 0
        dy = DA.from_host(np.full(n, 7.0), np.float64)
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
        assert np.array_equal(dy.to_host().view(np.uint64), yf.view(np.uint64))
    finally:
        check(lib.liship_spmv_csr_set_dom_march(1))
        check(lib.liship_csr_plan_destroy(plan))


def test_box27_marching_needs_a_box_of_shared_values(lib):
    """what plan time refuses: rows that do not carry the dominant pattern's values (one row with a value of its own: its pattern splits off), an infinite
    coefficient (its product with the halo's zero would be NaN), lines that 128 does not divide -- those plans keep the staged kernel and still give the oracle's bits"""
    cases = {}
    ptr, idx, val = stencil_box((9, 8, 128))
    v2 = val.copy()
    v2[ptr[500] + 2] = 3.0                                      # ONE row with a value of its own: the pattern splits, the rows no longer share the dominant values
    cases["one_row_differs"] = (ptr, idx, v2)
    v27 = -np.ones(27); v27[13] = 26.0; v27[5] = np.inf
    cases["infinite_coefficient"] = _box27_per_slot((9, 8, 128), v27)
    cases["lines_of_96"] = stencil_box((9, 8, 96))
    for name, (ptr, idx, val) in cases.items():
        n = len(ptr) - 1
        x = np.random.default_rng(4).uniform(-1, 1, n)
        dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
        check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
        check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
        check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
        try:
            check(lib.liship_spmv_csr_set_dom_march(2))
            assert lib.liship_csr_plan_box27(plan) == 0, name
            dy = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            assert np.array_equal(dy.to_host().view(np.uint64), orc.spmv_csr(ptr, idx, val, x).view(np.uint64)), name
        finally:
            check(lib.liship_spmv_csr_set_dom_march(1))
            check(lib.liship_csr_plan_destroy(plan))


def _blocked_rows(ptr, idx, val, bs, first_seen=True):
    """the CSR rows that list lis_matvec_bsr's terms of every scalar row of the bs x bs blocking (block after block, column after column, explicit zeros
    included): what liship_bsr_to_rows lays out in HBM for a constant-coefficient BSR matrix.  first_seen: a block row's blocks in the order
    lis_matrix_convert_csr2bsr meets them (lis_matrix_bsr.c:351-552: row after row of the block row, entry after entry) -- NOT ascending; else sorted."""
    import scipy.sparse as sp
    n = len(ptr) - 1
    assert n % bs == 0
    B = sp.csr_matrix((val, idx, ptr), shape=(n, n)).tobsr(blocksize=(bs, bs))
    B.sort_indices()
    nb = np.diff(B.indptr)
    rptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.repeat(nb, bs) * bs, out=rptr[1:])
    ridx, rval = np.empty(rptr[-1], np.int32), np.empty(rptr[-1])
    for i in range(n // bs):
        blocks = np.arange(B.indptr[i], B.indptr[i + 1])
        if first_seen:
            met = idx[ptr[i * bs]:ptr[(i + 1) * bs]] // bs
            _, first = np.unique(met, return_index=True)           # (sorted block columns = B.indices[blocks]; their first positions)
            blocks = blocks[np.argsort(first, kind="stable")]
        cols = (B.indices[blocks, None] * bs + np.arange(bs)[None, :]).ravel()
        for k in range(bs):
            r = i * bs + k
            ridx[rptr[r]:rptr[r + 1]] = cols
            rval[rptr[r]:rptr[r + 1]] = B.data[blocks, k, :].ravel()
    return rptr.astype(np.int32), ridx, rval


@pytest.mark.parametrize("case", ["p3d_2x2", "p2d_2x2", "p3d_2x2_long", "p3d_2x2_sorted", "p3d_3x3", "p3d_4x4"])
def test_rows_that_take_turns_on_several_patterns_share_a_virtual_dominant_one(lib, case):
    """the row form of a b x b blocked stencil has b interior patterns that take turns, none with half of the rows: the plan stages x for the UNION of the most
    frequent ones and every row is a mask over it with the union's values in scalar registers (build_wide_dominant).  Bits of the plain loop over the listed
    terms -- explicit zeros meet Inf and NaN as in lis_matvec_bsr.c:293-343, slots a row does not have add -0.0 -- whole, in row ranges, with the fused dots."""
    if case == "p2d_2x2":
        ptr, idx, val = orc.poisson3d(1, 48, 64, sort_cols=True)
    else:
        ptr, idx, val = orc.poisson3d(*((6, 6, 128) if case == "p3d_2x2_long" else (24, 24, 24)), sort_cols=True)
    bs = 3 if case == "p3d_3x3" else 4 if case == "p3d_4x4" else 2
    rptr, ridx, rval = _blocked_rows(ptr, idx, val, bs, first_seen=case != "p3d_2x2_sorted")
    n = len(rptr) - 1
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, n)
    x[::7] = 0.0
    x[3::11] = -0.0
    for pos, v in ((0, np.inf), (n - 1, -np.inf), (n // 2, np.nan), (n // 3, np.inf), (257, np.nan), (5, -np.inf)):
        x[pos] = v
    w = rng.uniform(-1, 1, n)
    ref = orc.spmv_csr(rptr, ridx, rval, x)
    nanpos = np.isnan(ref)
    dptr, didx, dval, dx, dw = DA.from_host(rptr, np.int32), DA.from_host(ridx, np.int32), DA.from_host(rval, np.float64), DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    taken = []
    try:
        for union in (2, 0):                              # (2: at any size; the default keeps it for plans of 2^19 rows and more)
            lib.liship_spmv_csr_set_wide_union(union)
            plan = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
            check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
            check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
            check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
            assert lib.liship_csr_plan_value_records(plan) == 2
            taken.append(lib.liship_csr_plan_wide_dominant(plan))
            if union == 0:                                  # ... and a lane per block row on top of the same records: switched on for the second plan, so that
                lib.liship_spmv_csr_set_block_rows(2)       # both the row-by-row kernels and this one are checked (2: at any size)
                check(lib.liship_csr_plan_encode_block_rows(plan, bs, dptr.ptr, None))
                assert lib.liship_csr_plan_block_rows(plan) == (0 if case == "p3d_2x2_long" else bs)      # (the 6 x 6 x 128 bar: no block row pattern with half of them)
            for variant in (0, 0x4000):
                lib.liship_spmv_csr_set_variant(variant)
                dy = DA.from_host(np.full(n, 7.0), np.float64)
                check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                y = dy.to_host()
                assert np.array_equal(np.isnan(y), nanpos), (union, hex(variant))
                assert np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), (union, hex(variant))
                for cut in (1, bs):                         # row ranges that cut block rows (the row-by-row kernels serve), and ranges of whole block rows
                    dy = DA.from_host(np.full(n, 7.0), np.float64)
                    a, b = n // 3 + 1, n - n // 5 - 3
                    a, b = a - a % cut, b - b % cut
                    for lo, hi in ((a, b), (0, a), (b, n)):
                        check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                    y = dy.to_host()
                    assert np.array_equal(np.isnan(y), nanpos) and np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), (union, "rows", cut, hex(variant))
            lib.liship_spmv_csr_set_variant(0)
            # the fused dots on finite data: y the same bits, the sums to rounding
            xf = np.where(np.isfinite(x), x, 0.5)
            dxf = DA.from_host(xf, np.float64)
            yf = orc.spmv_csr(rptr, ridx, rval, xf)
            res = DA.from_host(np.full(2, np.nan), np.float64)
            dy = DA.from_host(np.full(n, 7.0), np.float64)
            if lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dxf.ptr, dy.ptr, dw.ptr, 1, res.ptr, work.ptr, None) == 0:
                assert np.array_equal(dy.to_host().view(np.uint64), yf.view(np.uint64))
                np.testing.assert_allclose(res.to_host(), [np.dot(w, yf), np.dot(yf, yf)], rtol=1e-12)
            check(lib.liship_csr_plan_destroy(plan))
    finally:
        lib.liship_spmv_csr_set_wide_union(1)
        lib.liship_spmv_csr_set_block_rows(1)
        lib.liship_spmv_csr_set_variant(0)
    assert taken[1] == 0 and (taken[0] == 1 or bs > 2), taken      # (3 x 3, 4 x 4: the rows' turns have no common supersequence of 32 entries; their block rows do not need one)


@pytest.mark.parametrize("dims", [(9, 8, 128), (10, 16, 256), (8, 8, 256),
                                  (9, 10, 192), (10, 13, 132), (9, 8, 200), (8, 9, 254), (9, 17, 128)])      # partial tiles (round 5): lines that 128 does not divide, planes whose lines 8 does not
@pytest.mark.parametrize("order", ["first_seen_generator", "first_seen_sorted", "ascending"])
def test_block2_marching_kernel_bit_exact(lib, dims, order):
    """Round 5: the 7-point stencil kept as 2 x 2 blocks (Lis's default BSR block size, constant coefficients) on a box grid walks the planes like the scalar
    stencil does (spmv_csr_block2_march_kernel): a lane's pair of rows is a block row, 14 terms per row -- explicit zeros included, block after block in the order
    the conversion met them (lis_matvec_bsr.c:293-343) -- and a block outside the grid is a pair of zeros in the halo.  The bits of the plain loop over the listed
    terms: whole, in ranges of whole planes, with the fused dots, x with Inf / NaN / signed zeros (an explicit zero times Inf must stay NaN), three block orders."""
    nz, ny, nx = dims
    ptr, idx, val = orc.poisson3d(nz, ny, nx, sort_cols=(order != "first_seen_generator"))
    rptr, ridx, rval = _blocked_rows(ptr, idx, val, 2, first_seen=order != "ascending")
    n = len(rptr) - 1
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, n)
    x[::7] = 0.0
    x[3::11] = -0.0
    for pos, v in ((0, np.inf), (n - 1, -np.inf), (n // 2, np.nan), (n // 3, np.inf), (257, np.nan), (ny * nx + 5, -np.inf)):
        x[pos] = v
    w = rng.uniform(-1, 1, n)
    ref = orc.spmv_csr(rptr, ridx, rval, x)
    nanpos = np.isnan(ref)
    dptr, didx, dval, dx, dw = DA.from_host(rptr, np.int32), DA.from_host(ridx, np.int32), DA.from_host(rval, np.float64), DA.from_host(x, np.float64), DA.from_host(w, np.float64)
    work, res = DA(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
    mn = ny * nx
    try:
        lib.liship_spmv_csr_set_wide_union(0)
        lib.liship_spmv_csr_set_block_rows(2)
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
        check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
        check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
        check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
        check(lib.liship_csr_plan_encode_block_rows(plan, 2, dptr.ptr, None))
        assert lib.liship_csr_plan_block_rows(plan) == 2
        outs = {}
        for march in (2, 0):
            check(lib.liship_spmv_csr_set_dom_march(march))
            assert lib.liship_csr_plan_block2_march(plan) == (1 if march else 0)
            dy = DA.from_host(np.full(n, 7.0), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            y = dy.to_host()
            assert np.array_equal(np.isnan(y), nanpos) and np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), march
            dy = DA.from_host(np.full(n, 7.0), np.float64)
            for lo, hi in ((mn, n - mn), (0, mn), (n - mn, n)):
                check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            y = dy.to_host()
            assert np.array_equal(np.isnan(y), nanpos) and np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), (march, "planes")
            xf = np.where(np.isfinite(x), x, 0.5)
            dxf = DA.from_host(xf, np.float64)
            yf = orc.spmv_csr(rptr, ridx, rval, xf)
            for wname, wd, wh in (("x", dxf, xf), ("w", dw, w)):
                dy = DA.from_host(np.full(n, 7.0), np.float64)
                check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dxf.ptr, dy.ptr, wd.ptr, 1, res.ptr, work.ptr, None))
                assert np.array_equal(dy.to_host().view(np.uint64), yf.view(np.uint64)), (march, wname)
                got = res.to_host()
                assert abs(got[0] - np.dot(wh, yf)) <= 1e-12 * np.abs(wh * yf).sum() and abs(got[1] - np.dot(yf, yf)) <= 1e-12 * np.dot(yf, yf), (march, wname, got)
        check(lib.liship_csr_plan_destroy(plan))
    finally:
        check(lib.liship_spmv_csr_set_dom_march(1))
        lib.liship_spmv_csr_set_wide_union(1)
        lib.liship_spmv_csr_set_block_rows(1)


@pytest.mark.parametrize("case", ["constant", "ell_padded", "values_differ", "foreign_rows", "two_tiles_wide", "generator_order", "other_order", "slab_of_rank_0", "slab_of_rank_1", "dia_zeros",
                                  # lines that are not a multiple of 128 long (round 5): the last tile of a line is partial -- 64, 4, 72, 126, 8 and 60 columns wide
                                  "constant@192", "ell_padded@132", "dia_zeros@200", "generator_order@254", "slab_of_rank_1@136", "foreign_rows@188", "values_differ@320", "other_order@192",
                                  # ... planes whose lines are not a multiple of 8 (the last tile of a plane is partial: 1 .. 7 lines), lines shorter than a tile
                                  "constant@128x9", "constant@192x15", "ell_padded@132x12", "dia_zeros@100x13", "generator_order@64x10", "slab_of_rank_0@200x11", "slab_of_rank_1@72x17",
                                  "foreign_rows@100x20", "values_differ@140x14", "other_order@96x9"])
def test_z_marching_form_of_the_dominant_pattern_product(lib, case):
    """the 7-point stencil with value records on a grid whose lines are a multiple of 128 long: a workgroup walks the planes of its 128 x 8 tile, each x loaded once
    (spmv_csr_valuerec_march_kernel; 512^3: 0.49 -> 0.41 ms).  The oracle's bits with the form on and off -- Inf / NaN / -0.0 in x, rows of other patterns (faces: masks;
    other values: the waterfall's; foreign rows: their own records; ELL's padding terms), whole launches, plane ranges (the form) and ranges that cut planes (the
    gathering kernel), the fused dots."""
    case, _, line = case.partition("@")
    nz, ny, nx = (12, 16, 256) if case == "two_tiles_wide" else (20, 16, 128)
    if line:
        nx = int(line.split("x")[0])
        ny = int(line.split("x")[1]) if "x" in line else ny
    ptr, idx, val = orc.poisson3d(nz, ny, nx, sort_cols=True)
    n = len(ptr) - 1
    SO = ny * nx
    ncols = n
    foreign_planes = set()
    if case == "slab_of_rank_0":                              # the first nz planes of a taller grid: the last plane's rows are on the DOMINANT pattern, their +SO neighbours
        ptr, idx, val = orc.poisson3d(nz + 3, ny, nx, sort_cols=True, is_=0, ie=n)      # beyond the rows (a multi-rank job's ghost columns right behind x[0, n))
        ncols = n + SO
    if case == "slab_of_rank_1":                              # planes 2 .. nz + 1 of a taller grid with the ghost planes renumbered behind the rows, lower neighbour first
        ptr, idx, val = orc.poisson3d(nz + 4, ny, nx, sort_cols=True, is_=2 * SO, ie=2 * SO + n)      # (as lis_matrix_g2l does): the first and last planes' rows are foreign
        idx = idx.astype(np.int64) - 2 * SO
        idx = np.where(idx < 0, n + (idx + SO), np.where(idx >= n, n + SO + (idx - n), idx)).astype(np.int32)
        ncols = n + 2 * SO
    rng = np.random.default_rng(11)
    if case == "ell_padded":                                  # every row padded to 7 entries with (row, +0.0) behind its own, as lis_matrix_convert_csr2ell lays them out
        p0, i0, v0 = ptr, idx, val
        ptr = np.arange(0, 7 * n + 1, 7, dtype=np.int32)
        idx, val = np.empty(7 * n, np.int32), np.zeros(7 * n)
        lens = np.diff(p0)
        for r in np.flatnonzero(lens < 7):
            k = lens[r]
            idx[7 * r:7 * r + k], val[7 * r:7 * r + k] = i0[p0[r]:p0[r + 1]], v0[p0[r]:p0[r + 1]]
            idx[7 * r + k:7 * r + 7] = r
        full = np.flatnonzero(lens == 7)
        idx.reshape(n, 7)[full] = i0[(p0[full][:, None] + np.arange(7)[None, :])]
        val.reshape(n, 7)[full] = v0[(p0[full][:, None] + np.arange(7)[None, :])]
    if case == "dia_zeros":                                   # DIA's row form (liship_dia_to_rows): every diagonal that stays inside the array, explicit +0.0 where the grid has no neighbour
        offs = np.array([-SO, -nx, -1, 0, 1, nx, SO])
        rows = np.arange(n)
        cols = rows[:, None] + offs[None, :]
        inside = (cols >= 0) & (cols < n)
        import scipy.sparse as sp
        A = sp.csr_matrix((val, idx, ptr), shape=(n, n))
        vals = np.zeros(cols.shape)
        vals[inside] = np.asarray(A[np.repeat(rows, 7).reshape(n, 7)[inside], cols[inside]]).ravel()
        ptr = np.concatenate(([0], np.cumsum(inside.sum(1)))).astype(np.int32)
        idx, val = cols[inside].astype(np.int32), vals[inside]
    if case == "values_differ":                               # the rows of three planes carry another diagonal: the same offsets, values of their own
        rows = np.repeat(np.arange(n), np.diff(ptr))
        val = val.copy()
        val[(idx == rows) & (rows // SO % 5 == 2)] = 7.5
    if case == "foreign_rows":                                # a few interior rows whose +1 neighbour is the +5 one instead: seven entries, not a subsequence of the dominant pattern
        idx, val = idx.copy(), val.copy()
        for r in np.sort(rng.choice(np.arange(3 * SO, 9 * SO), 40, replace=False)):
            k = ptr[r] + int(np.flatnonzero(idx[ptr[r]:ptr[r + 1]] == r + 1)[0]) if (idx[ptr[r]:ptr[r + 1]] == r + 1).any() else -1
            if k >= 0 and r + 5 not in idx[ptr[r]:ptr[r + 1]]:
                idx[k], val[k] = r + 5, 0.25
                foreign_planes.add(int(r // SO))
    if case == "generator_order":                             # the slot order of the reference's generators (-SO, +SO, -S, +S, -1, +1, 0)
        ptr, idx, val = orc.poisson3d(nz, ny, nx, sort_cols=False)
    if case == "other_order":                                 # some other order of the same seven columns: the kernel's general form
        idx, val = idx.copy(), val.copy()
        full = np.flatnonzero(np.diff(ptr) == 7)
        sel = np.array([3, 0, 6, 2, 4, 1, 5])
        I, V = idx[ptr[full][:, None] + sel[None, :]], val[ptr[full][:, None] + sel[None, :]]
        idx[ptr[full][:, None] + np.arange(7)[None, :]], val[ptr[full][:, None] + np.arange(7)[None, :]] = I, V
    x = rng.uniform(-1, 1, ncols)
    x[::7] = 0.0
    x[3::11] = -0.0
    for pos, v in ((0, np.inf), (n - 1, -np.inf), (n // 2, np.nan), (n // 3, np.inf), (SO + 129, np.nan), (5, -np.inf), (ncols - 3, np.inf)):
        x[pos] = v
    ref = orc.spmv_csr(ptr, idx, val, x)
    nanpos = np.isnan(ref)
    xf = np.where(np.isfinite(x), x, 0.5)
    yf = orc.spmv_csr(ptr, idx, val, xf)
    w = rng.uniform(-1, 1, n)
    dptr, didx, dval, dx, dxf, dw = (DA.from_host(a, t) for a, t in ((ptr, np.int32), (idx, np.int32), (val, np.float64), (x, np.float64), (xf, np.float64), (w, np.float64)))
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    assert lib.liship_csr_plan_dominant_pattern(plan) == 1
    # a box (no pattern bytes, no masks: signed zeros in the halo) where every face row is the dominant pattern minus the neighbours outside the grid
    # a box (no pattern bytes, no masks: signed zeros in the halo) in the planes where every row is the dominant pattern minus the neighbours outside the grid, with its values
    bad = {"other_order": set(range(nz)),                     # ("other_order" permutes the full rows only: its face rows are foreign; ELL's padding terms and DIA's zeros are box forms of their own)
           "dia_zeros": {0, nz - 1},                          # (the first and the last plane lose the diagonals that leave the array: another rule than the planes between)
           "values_differ": {z for z in range(nz) if z % 5 == 2}, "foreign_rows": foreign_planes,
           "slab_of_rank_1": {0, nz - 1}}.get(case, set())      # (rank 0's last plane keeps its +SO slot, the ghost plane behind it: a box whose far side is "there, with a real x")
    runs = "".join("x" if z in bad else "o" for z in range(nz)).split("x")
    assert lib.liship_csr_plan_box_planes(plan) == max(len(r) for r in runs), (case, sorted(bad))
    sums = []
    try:
        for march in (2, 3, 0):                               # 2: at any size (the default leaves grids this small to the gathering kernel); 3: the masks' form on a box too
            lib.liship_spmv_csr_set_dom_march(march)
            if line and march:
                assert lib.liship_csr_plan_marching(plan) >= 1, (case, line)      # the partial tile is served by the marching kernel, not by a fall-back
            dy = DA.from_host(np.full(n, 7.0), np.float64)
            check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
            y = dy.to_host()
            assert np.array_equal(np.isnan(y), nanpos), march
            assert np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), march
            for ranges in (((0, 9 * SO), (9 * SO, n)), ((0, 3 * SO + 77), (3 * SO + 77, 12 * SO), (12 * SO, n)) if nz > 12 else ((0, 5 * SO), (5 * SO, n))):
                dy = DA.from_host(np.full(n, 7.0), np.float64)
                for lo, hi in ranges:
                    check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                y = dy.to_host()
                assert np.array_equal(np.isnan(y), nanpos) and np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), (march, ranges)
            for wv, want_w in ((dxf, xf), (dw, w)):           # w = x (the diagonal's pair serves) and a vector of its own
                for sumsq in (0, 1):
                    res = DA.from_host(np.full(2, np.nan), np.float64)
                    dy = DA.from_host(np.full(n, 7.0), np.float64)
                    check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dxf.ptr, dy.ptr, wv.ptr, sumsq, res.ptr, work.ptr, None))
                    assert np.array_equal(dy.to_host().view(np.uint64), yf.view(np.uint64)), (march, sumsq)
                    got = res.to_host()
                    np.testing.assert_allclose(got[0], np.dot(want_w[:n], yf), rtol=1e-12)
                    if sumsq:
                        np.testing.assert_allclose(got[1], np.dot(yf, yf), rtol=1e-12)
                    sums.append(got.copy())
    finally:
        lib.liship_spmv_csr_set_dom_march(1)
        check(lib.liship_csr_plan_destroy(plan))


@pytest.mark.parametrize("seed", range(int(os.environ.get("LIS_AMD_FUZZ_SEEDS", "60")) // 2))
def test_z_marching_on_perturbed_grids(lib, seed):
    """random 7-point grids (lines of 128 .. 322 columns), in CSR (either slot order), ELL's padded or DIA's zero-filled row form, with random planes perturbed --
    a value changed, an entry of an interior row removed, a column moved -- so that the box the plan finds is some run of planes between them: every form of the
    product (marching at any size, its masks' form, the gathering kernel), whole and in ranges that cut planes, must give the oracle's bits"""
    rng = np.random.default_rng(9000 + seed)
    nx, ny, nz = int(rng.choice([128, 256, 132, 190, 200, 254, 322, 64, 100])), int(rng.choice([8, 16, 24, 9, 13, 20, 31])), int(rng.integers(9, 22))      # (partial last tiles in x and y too: round 5)
    flavour = ["csr", "csr_generator", "ell", "dia"][seed % 4]
    ptr, idx, val = orc.poisson3d(nz, ny, nx, sort_cols=flavour != "csr_generator")
    n, SO = len(ptr) - 1, ny * nx
    ptr = ptr.astype(np.int64)
    rows_of = np.repeat(np.arange(n), np.diff(ptr))
    idx, val = idx.astype(np.int64), val.copy()
    keep = np.ones(len(idx), bool)
    for z in rng.choice(nz, int(rng.integers(0, 3)), replace=False):
        r = int(z) * SO + int(rng.integers(0, SO))
        kind = int(rng.integers(0, 3))
        ks = np.arange(ptr[r], ptr[r + 1])
        off = ks[idx[ks] != r]
        if kind == 0:
            val[rng.choice(ks)] = 0.375                       # a value of its own
        elif kind == 1 and len(off):
            keep[rng.choice(off)] = False                     # an entry missing inside the grid
        elif len(off):
            k = int(rng.choice(off))
            c = int(np.clip(idx[k] + 3, 0, n - 1))
            if c not in idx[ks]:
                idx[k] = c                                    # a column of its own
    idx, val, rows_of = idx[keep], val[keep], rows_of[keep]
    ptr = np.concatenate(([0], np.cumsum(np.bincount(rows_of, minlength=n))))
    if flavour == "ell":                                      # rows padded to the longest with (row, +0.0)
        w = int(np.diff(ptr).max())
        I = np.repeat(np.arange(n), w).reshape(n, w)
        V = np.zeros((n, w))
        for r in np.flatnonzero(np.diff(ptr) < w):
            k = ptr[r + 1] - ptr[r]
            I[r, :k], V[r, :k] = idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]]
        full = np.flatnonzero(np.diff(ptr) == w)
        I[full] = idx[ptr[full][:, None] + np.arange(w)[None, :]]
        V[full] = val[ptr[full][:, None] + np.arange(w)[None, :]]
        ptr, idx, val = np.arange(0, n * w + 1, w), I.ravel(), V.ravel()
    if flavour == "dia":                                      # every diagonal of the matrix that stays inside the array, explicit +0.0
        import scipy.sparse as sp
        A = sp.csr_matrix((val, idx, ptr), shape=(n, n))
        offs = np.unique(idx - rows_of)
        cols = np.arange(n)[:, None] + offs[None, :]
        inside = (cols >= 0) & (cols < n)
        vals = np.zeros(cols.shape)
        vals[inside] = np.asarray(A[np.repeat(np.arange(n), len(offs)).reshape(n, -1)[inside], cols[inside]]).ravel()
        ptr, idx, val = np.concatenate(([0], np.cumsum(inside.sum(1)))), cols[inside], vals[inside]
    ptr, idx = ptr.astype(np.int32), idx.astype(np.int32)
    x = rng.uniform(-1, 1, n)
    x[rng.integers(0, n, 6)] = [np.inf, -np.inf, np.nan, 0.0, -0.0, np.inf]
    ref = orc.spmv_csr(ptr, idx, val, x)
    nanpos = np.isnan(ref)
    dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    state = (flavour, nx, ny, nz, lib.liship_csr_plan_value_records(plan), lib.liship_csr_plan_dominant_pattern(plan), lib.liship_csr_plan_box_planes(plan))
    print("marching fuzz", seed, state)
    try:
        for march in (2, 3, 0):
            lib.liship_spmv_csr_set_dom_march(march)
            cut = int(rng.integers(1, nz - 1)) * SO + int(rng.integers(0, 2)) * 77
            for ranges in (((0, n),), ((0, cut), (cut, n))):
                dy = DA.from_host(np.full(n, 7.0), np.float64)
                for lo, hi in ranges:
                    check(lib.liship_spmv_csr_rows_f64(plan, lo, hi, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, None))
                y = dy.to_host()
                assert np.array_equal(np.isnan(y), nanpos), (seed, state, march, ranges)
                assert np.array_equal(y[~nanpos].view(np.uint64), ref[~nanpos].view(np.uint64)), (seed, state, march, ranges)
    finally:
        lib.liship_spmv_csr_set_dom_march(1)
        check(lib.liship_csr_plan_destroy(plan))
