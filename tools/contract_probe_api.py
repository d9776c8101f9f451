"""The contract form through lis_matvec (as bench.py's contract_form leg runs it), HIP-event ms per launch:   python tools/contract_probe_api.py [N=512] [launches=30]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
from lis_amd import _capi as capi, check  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = lis_amd.load()
dll = lib.dll
assert lib.initialize([]) == 0
dll.lis_amd_set_residency(1)
dll.lis_amd_stream.restype = C.c_void_p
A = capi.PM()
assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, 0, N ** 3) == 0
dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
assert dll.lis_amd_matrix_poisson3d(A, N, N, N, 0) == 0
n, nnz = A.contents.n, A.contents.nnz
x, xg, y = capi.PV(), capi.PV(), capi.PV()
for v in (x, xg, y):
    assert lib.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(v)) == 0
assert lib.lis_vector_set_all(1.0, x) == 0
chunk = 1 << 24
for s0 in range(0, n, chunk):
    cnt = min(chunk, n - s0)
    part = np.modf(np.arange(s0, s0 + cnt, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, s0, cnt, part.ctypes.data_as(capi.P_DBL), xg) == 0
stream = dll.lis_amd_stream()
timer, ev = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))
alg = 12 * nnz + 20 * n + 4


def measure(tag, xv, bytes_):
    for _ in range(10):
        assert lib.lis_matvec(A, xv, y) == 0
    check(lib.liship_timer_start(timer, stream))
    for _ in range(launches):
        assert lib.lis_matvec(A, xv, y) == 0
    check(lib.liship_timer_stop(timer, stream))
    dll.lis_amd_synchronize()
    check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
    ms = ev.value / launches
    print(f"{tag:40s} {ms:7.4f} ms  frac={bytes_ / ms / 1e6 / 8000:.4f}", flush=True)


for rep in range(3):
    measure("default (marching), x=1", x, 16 * n)
    check(lib.liship_spmv_csr_set_row_values(0))
    measure("values streamed (pattern7), x=1", x, 8 * nnz + 17 * n + 4)
    measure("values streamed (pattern7), xg", xg, 8 * nnz + 17 * n + 4)
    check(lib.liship_spmv_csr_set_row_patterns(0)); check(lib.liship_spmv_csr_set_index_codes(0))
    for strips in (1, 0):
        check(lib.liship_spmv_csr_set_xcd_strips(strips))
        measure(f"contract form, x=1, strips={strips}", x, alg)
        measure(f"contract form, xg, strips={strips}", xg, alg)
    check(lib.liship_spmv_csr_set_xcd_strips(1))
    check(lib.liship_spmv_csr_set_row_patterns(1)); check(lib.liship_spmv_csr_set_index_codes(1)); check(lib.liship_spmv_csr_set_row_values(1))
