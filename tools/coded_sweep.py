"""geometry sweep of the coded-index CSR kernel on the 512^3 stencil.   python tools/coded_sweep.py [N]"""
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
x, y = DA(n, np.float64), DA(n, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
bytes_alg = 12 * nnz + 20 * n + 4
for geom in (1,):
    for extra in (0, 0x100, 0x020001, 0x030001, 0x040001, 0x080001, 0x100001, 0x400001, 0):                         # 0x100: ablation without the x gather (wrong results, timing only)
        variant = (geom << 4)
        lib.liship_spmv_csr_set_variant(variant)
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
        check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
        lib.liship_spmv_csr_set_variant(variant | extra)
        usel = extra
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)))
        print(f"geom {geom} usel {usel} coded={lib.liship_csr_plan_coded(plan)}: {ms:.4f} ms  variant {variant | extra:#x}  {2 * nnz / ms / 1e6:.1f} GFLOP/s  {bytes_alg / ms / 1e6 / 80:.1f}%", flush=True)
        lib.liship_csr_plan_destroy(plan)
lib.liship_spmv_csr_set_variant(0)
