"""The constant-coefficient products in one and two dimensions (the reference's spmvtest1 / spmvtest2 matrices) at HBM sizes: python tools/lowdim_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402

lib = lis_amd.load()
for name, (L, M, N) in (("1-D 3-point, n = 2^27", (1, 1, 1 << 27)), ("2-D 5-point, 8192 x 16384", (1, 8192, 16384)), ("2-D 5-point, 11584^2", (1, 11584, 11584)), ("3-D 7-point 512^3", (512, 512, 512))):
    n = L * M * N
    nnz = lib.liship_poisson3d_nnz(L, M, N, 0, n)
    dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
    x, y = DA(n, np.float64), DA(n, np.float64)
    check(lib.liship_poisson3d_csr(L, M, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
    check(lib.liship_memset(x.ptr, 0, 8 * n, None))
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=30, warm=10)
    print(f"{name:28s} n {n:10d} nnz {nnz:11d}: value records {lib.liship_csr_plan_value_records(plan)} dominant {lib.liship_csr_plan_dominant_pattern(plan)} marching {lib.liship_csr_plan_marching(plan)}: "
          f"{ms:.4f} ms  {2e-6 * nnz / ms:.0f} GFLOP/s  {17e-6 * n / ms / 8000:.3f} of 8 TB/s on 17 B/row", flush=True)
    check(lib.liship_csr_plan_destroy(plan))
    for a in (dptr, didx, dval, x, y):
        a.free()
