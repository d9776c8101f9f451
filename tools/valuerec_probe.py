"""Value records A/B at N^3 Poisson: python tools/valuerec_probe.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402

lib = lis_amd.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
x, y, y2 = DA(n, np.float64), DA(n, np.float64), DA(n, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
x.upload(np.cos(0.01 * np.arange(n)) + 1.25)
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
print("row patterns:", lib.liship_csr_plan_row_patterns(plan), "value records:", lib.liship_csr_plan_value_records(plan), flush=True)
for rep in range(3):
    for on in (1, 0):
        lib.liship_spmv_csr_set_row_values(on)
        yy = y if on else y2
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, yy.ptr, None)), iters=50, warm=20)
        print(f"value records {on}: {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s", flush=True)
lib.liship_spmv_csr_set_row_values(1)
a, b = y.to_host(), y2.to_host()
print("bit-identical:", bool(np.array_equal(a.view(np.uint64), b.view(np.uint64))), flush=True)
