"""XCD-run block order of the coded CSR kernel under the fabric counters: python tools/xrun_probe.py <variant-hex> [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402

lib = lis_amd.load()
variant = int(sys.argv[1], 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
x, y = DA(n, np.float64), DA(n, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
lib.liship_spmv_csr_set_variant(0x10)
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
lib.liship_spmv_csr_set_variant(variant)
ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=30, warm=30)
print(f"variant {variant:#x}: {ms:.4f} ms", flush=True)
