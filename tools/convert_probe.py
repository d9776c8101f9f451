"""lis_matrix_convert at N^3 (7-point stencil, sorted rows), in HBM and on the host arrays: python tools/convert_probe.py [N]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd, lisdrv, orc
from lis_amd import _capi as capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = lis_amd.load(); assert lib.initialize([]) == 0
dll = lib.dll
dll.lis_amd_matrix_lazy_arrays.argtypes = [capi.PM]
ptr, idx, val = orc.poisson3d(N, N, N, sort_cols=True)
n = N ** 3
A = lisdrv.make_csr(lib, ptr, idx, val)
x, y = lisdrv.new_vector(lib, A, np.ones(n)), lisdrv.new_vector(lib, A)
assert lib.lis_matvec(A, x, y) == 0
for fmt, bs in (("ell", 0), ("dia", 0), ("csc", 0), ("bsr", 2), ("jad", 0)):
    for where in ("hbm", "host"):
        dll.lis_amd_set_device_convert(1 if where == "hbm" else 0)
        dll.lis_amd_synchronize()
        t0 = time.perf_counter()
        B = lisdrv.convert(lib, A, fmt, bs or 2, bs or 2)
        dll.lis_amd_synchronize()
        t1 = time.perf_counter()
        assert lib.lis_matvec(B, x, y) == 0
        dll.lis_amd_synchronize()
        t2 = time.perf_counter()
        nrm = C.c_double(); lib.lis_vector_nrm2(y, C.byref(nrm))
        print(f"{fmt:4s} {where:5s} convert {1e3 * (t1 - t0):9.2f} ms   first product {1e3 * (t2 - t1):8.2f} ms   host arrays still in HBM only: {dll.lis_amd_matrix_lazy_arrays(B)}   ||A*1|| = {nrm.value:.6e}", flush=True)
        lib.lis_matrix_destroy(B)
dll.lis_amd_set_device_convert(1)
