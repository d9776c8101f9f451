"""Setup cost of a 512^3 CSR matrix in HBM: generation + row split, with and without index coding.  python tools/plan_time.py [N]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
from lis_amd import _capi as capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
lib.dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
lib.dll.lis_amd_matrix_index_codes.argtypes = [capi.PM]
for rep in range(3):
    A = capi.PM()
    assert lib.lis_matrix_create(0, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, 0, N ** 3) == 0
    lib.dll.lis_amd_synchronize()
    t0 = time.perf_counter()
    assert lib.dll.lis_amd_matrix_poisson3d(A, N, N, N, 0) == 0
    lib.dll.lis_amd_synchronize()
    dt = time.perf_counter() - t0
    print(f"N={N} rep {rep}: generate + plan{' + encode' if not os.environ.get('LIS_AMD_NO_INDEX_CODES') else ''}: {dt * 1e3:.1f} ms, codes={lib.dll.lis_amd_matrix_index_codes(A)}", flush=True)
    lib.lis_matrix_destroy(A)
