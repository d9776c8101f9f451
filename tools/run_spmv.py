"""Run the CSR SpMV a few times on one cubic Poisson grid (profiling target for rocprofv3).

    python tools/run_spmv.py N [variant] [iters] [sorted]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402

lib = lis_amd.load()
N = int(sys.argv[1])
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sorted_ = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
x, y = DA(n, np.float64), DA(n, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, sorted_, dptr.ptr, didx.ptr, dval.ptr, None))
check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
lib.liship_spmv_csr_set_variant(variant)
for _ in range(iters):
    check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))
check(lib.liship_device_synchronize())
print("done", N, variant, iters, "alg bytes", 12 * nnz + 20 * n + 4)
