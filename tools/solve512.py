"""Time to solution at BASELINE's full size: the 512^3 Poisson system (b = A*1, x0 = 0, tol 1e-12) through lis_solve
on one GPU, HBM-generated matrix, resident objects.   python tools/solve512.py [N]"""
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
for opts in ("-i cg -p jacobi", "-i cg -p none", "-i bicgstab -p none", "-i bicg -p none"):     # GMRES(30) stagnates on this system
    S = capi.PS()
    lib.lis_solver_create(C.byref(S))
    lib.lis_solver_set_option((opts + " -tol 1e-12 -maxiter 20000").encode(), S)
    assert lib.lis_solve(A, b, x, S) == 0
    s = S.contents
    print(f"N={N} {opts}: {s.iter} iterations, status {s.retcode}, rel. residual {s.resid:.3e}, {s.itime:.2f} s, {s.iter / s.itime:.1f} it/s", flush=True)
    lib.lis_solver_destroy(S)
