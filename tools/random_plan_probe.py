"""Plan-time cost of the renumbering ATTEMPT on matrices it cannot help (random columns): assemble + upload + plan, with and without LIS_AMD_NO_REORDER=1."""
import sys, os, time, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, lis_amd, lisdrv, orc
from lis_amd import _capi as capi
lib = lis_amd.load(); assert lib.initialize([]) == 0
dll = lib.dll
dll.lis_amd_matrix_reordered.argtypes = [capi.PM]; dll.lis_amd_matrix_reordered.restype = C.c_longlong
for rows, per in ((300000, 8), (300000, 40), (2000000, 8)):
    ptr, idx, val = orc.random_csr(rows, per, seed=3, ncols=rows, empty_rows=False)
    t0 = time.time(); A = lisdrv.make_csr(lib, ptr, idx, val); re = dll.lis_amd_matrix_reordered(A); t = time.time() - t0
    x = np.random.default_rng(1).uniform(-1, 1, rows)
    y = lisdrv.matvec(lib, A, x)
    ok = np.array_equal(y, orc.spmv_csr(ptr, idx, val, x))
    print(f"random {rows} x {per}: upload + plan {t:.2f} s, reordered form {re}, bits ok {ok}", flush=True)
    lib.lis_matrix_destroy(A)
