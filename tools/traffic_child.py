"""A few products of the 512^3 (N^3) Poisson matrix through lis_matvec, exactly as bench.py sets them up -- the process rocprofv3 profiles when bench.py measures
its `roofline.traffic` live (bench.py live_traffic):   python tools/traffic_child.py N form iters
form: 1 the plan's own choice, 0 value records off (values streamed), 2 the reference layout mode (lis_amd_set_reference_layout: 4 B indices + 8 B values streamed) on the non-trivial x"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
from lis_amd import _capi as capi, check  # noqa: E402

N, values, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
A = capi.PM()
assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, 0, N ** 3) == 0
lib.dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
assert lib.dll.lis_amd_matrix_poisson3d(A, N, N, N, 0) == 0
x, y = capi.PV(), capi.PV()
for v in (x, y):
    assert lib.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(v)) == 0
assert lib.lis_vector_set_all(1.0, x) == 0
if values == 0:
    check(lib.liship_spmv_csr_set_row_values(0))
if values == 2:
    assert lib.dll.lis_amd_set_reference_layout(1) == 0       # the mode bench.py's headline runs in, applied to the plan already built -- as bench.py does
x1 = x
if values == 2:                                               # ... on the headline's non-trivial x
    import numpy as np
    x = capi.PV()
    assert lib.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(x)) == 0
    n = N ** 3
    for s0 in range(0, n, 1 << 24):
        cnt = min(1 << 24, n - s0)
        part = np.modf(np.arange(s0, s0 + cnt, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
        assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, s0, cnt, part.ctypes.data_as(capi.P_DBL), x) == 0
for _ in range(iters):
    assert lib.lis_matvec(A, x, y) == 0
lib.dll.lis_amd_synchronize()
print("done", N, values, iters, flush=True)
