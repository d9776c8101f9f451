"""27-point stencil (constant coefficients: 26 on the diagonal, -1 on the 26 neighbours, Dirichlet truncation), CSR product through the plan:
python tools/stencil27_probe.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402


def stencil27(N):
    n = N ** 3
    z, y, x = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    z, y, x = z.ravel(), y.ravel(), x.ravel()
    cols, vals, rows = [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                m = (z + dz >= 0) & (z + dz < N) & (y + dy >= 0) & (y + dy < N) & (x + dx >= 0) & (x + dx < N)
                r = np.nonzero(m)[0]
                rows.append(r)
                cols.append(r + (dz * N + dy) * N + dx)
                vals.append(np.full(len(r), 26.0 if (dz, dy, dx) == (0, 0, 0) else -1.0))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((cols, rows))
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ptr[1:])
    return ptr.astype(np.int32), cols[order].astype(np.int32), vals[order]


lib = lis_amd.load()
import os as _os
if _os.environ.get('NO_XCD_STRIPS') == '1':
    lib.liship_spmv_csr_set_xcd_strips(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
ptr, idx, val = stencil27(N)
if "--varying" in sys.argv:                       # a non-uniform mesh: the pattern stays, every row has its own values
    val = np.random.default_rng(5).uniform(-1, 1, len(val))
n, nnz = len(ptr) - 1, len(idx)
dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
x, y = DA.from_host(np.cos(0.01 * np.arange(n)) + 1.25, np.float64), DA(n, np.float64)
yref = None
if N <= 200:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import orc
    yref = orc.spmv_csr(ptr, idx, val, np.cos(0.01 * np.arange(n)) + 1.25)
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
print(f"N={N} n={n} nnz={nnz}: codes {lib.liship_csr_plan_coded(plan)}, patterns {lib.liship_csr_plan_row_patterns(plan)}, records {lib.liship_csr_plan_pattern_records(plan)}, value records {lib.liship_csr_plan_value_records(plan)}", flush=True)
for codes, pats, vals in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
    lib.liship_spmv_csr_set_index_codes(codes)
    lib.liship_spmv_csr_set_row_patterns(pats)
    lib.liship_spmv_csr_set_row_values(vals)
    for variant in ((0, 0x4000, 0x2000, 0) if (codes, pats, vals) == (1, 1, 0) else (0, 0x4000, 0) if (codes, pats, vals) == (1, 1, 1) else (0,)):      # 0x2000: the general (one lane per row) pattern kernel
        lib.liship_spmv_csr_set_variant(variant)
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=50, warm=20)
        if yref is not None:
            assert np.array_equal(y.to_host(), yref), (codes, pats, vals, variant)
        own = 8 * nnz + 17 * n
        if (codes, pats, vals) == (1, 1, 1):
            print(f"  variant {variant:#x}: {ms:.4f} ms = {2e-6 * nnz / ms:.0f} GFLOP/s, {17e-9 * n / ms * 1e3 / 8000:.3f} of 8 TB/s on 17 B per row (wide dominant {lib.liship_csr_plan_wide_dominant(plan)}; 0x4000: a gather per entry)", flush=True)
        elif variant or (codes, pats, vals) == (1, 1, 0):
            print(f"  variant {variant:#x}: {ms:.4f} ms = {own / ms / 8e7:.1f} % of 8 TB/s on the 8 B per non-zero + 17 B per row it streams (0: four lanes per row, x staged; 0x4000: four lanes, gathers; 0x2000: round 2)", flush=True)
    lib.liship_spmv_csr_set_variant(0)
    alg = 12 * nnz + 20 * n
    print(f"codes {codes} patterns {pats} value records {vals}: {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s  {alg / ms / 1e6:.0f} GB/s on the contract's bytes ({alg / ms / 8e7:.1f} % of 8 TB/s)", flush=True)
lib.liship_spmv_csr_set_index_codes(1)
lib.liship_spmv_csr_set_row_patterns(1)
lib.liship_spmv_csr_set_row_values(1)
