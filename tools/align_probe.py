"""Does the product's time depend on where y sits relative to x?  python tools/align_probe.py [N]
one allocation [x | pad | y], pads from 0 to 1 MiB (+ odd multiples of the sizes a channel hash could use)"""
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
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
big = DA(2 * n + (4 << 20) // 8, np.float64)
x = DA(n, np.float64)
x.upload(np.ones(n))
check(lib.liship_memcpy_d2d(big.ptr, x.ptr, 8 * n, None))
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
xp = big.ptr if isinstance(big.ptr, int) else big.ptr.value
print(f"x at {xp:#x} (mod 2 MiB: {xp % (2 << 20):#x})", flush=True)
for rep in range(2):
    for pad in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 4096 + 256, 65536 + 4096):
        yp = xp + 8 * n + pad
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, xp, yp, None)), iters=40, warm=10)
        print(f"pad {pad:8d}: y - x = {(yp - xp) % (2 << 20):#9x} mod 2 MiB   {ms:.4f} ms", flush=True)
