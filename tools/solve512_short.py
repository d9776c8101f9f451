"""120 iterations of CG + Jacobi on the 512^3 Poisson system through lis_solve (profiling runs: tools/run_inloop_pmc.sh)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
import lisdrv   # noqa: E402
from lis_amd import _capi as capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
A = capi.PM()
assert lib.lis_matrix_create(0, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, 0, N ** 3) == 0
lib.dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
assert lib.dll.lis_amd_matrix_poisson3d(A, N, N, N, 0) == 0
b, x = lisdrv.new_vector(lib, A), lisdrv.new_vector(lib, A)
lib.dll.lis_amd_vector_poisson3d_rhs.argtypes = [capi.PV, C.c_int, C.c_int, C.c_int]
assert lib.dll.lis_amd_vector_poisson3d_rhs(b, N, N, N) == 0
S = capi.PS()
lib.lis_solver_create(C.byref(S))
lib.lis_solver_set_option(b"-i cg -p jacobi -tol 1e-30 -maxiter 120", S)
assert lib.lis_solve(A, b, x, S) == 0
print(f"N={N}: {S.contents.iter} iterations, {min(S.contents.iter, 120) / S.contents.itime:.1f} it/s", flush=True)
