"""Krylov iterations per second on the 27-point stencil with varying coefficients (values streamed), through lis_solve:
python tools/stencil27_solve_probe.py [N]     -- the fused product of the loops is the four-lanes-per-row product + the row blocks' sums
(variant 0) or the one-lane-per-row pattern kernel with its epilogue (variant 0x2000)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import lis_amd, lisdrv
from test_kernels_gpu import stencil_box
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 160
lib = lis_amd.load(); assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
ptr, idx, val = stencil_box((N, N, N))
n = len(ptr) - 1
rng = np.random.default_rng(5)
if "--constant" not in sys.argv:                     # (--constant: the reference's spmvtest3b / HPCG matrix -- wide value records, the staged kernel with the dominant pattern in scalar registers)
    val = val * rng.uniform(0.5, 1.5, len(val))      # still diagonally dominant in most rows; the pattern stays, every row has its own values
A = lisdrv.make_csr(lib, ptr, idx, val)
bb = rng.uniform(-1, 1, n)
for opts in ("-i gmres -restart 30 -p none", "-i bicgstab -p none", "-i cg -p jacobi"):
    for variant in ((0, 0x4000, 0) if "--constant" in sys.argv else (0, 0x2000, 0)):
        lib.liship_spmv_csr_set_variant(variant)
        out = lisdrv.solve(lib, A, bb, opts + " -maxiter 200 -tol 1e-30")
        print(f"{opts}: variant {variant:#x}: {out['iter']} iterations, {out['iter'] / out['itime']:.1f} it/s, residual {out['resid']:.3e}", flush=True)
lib.liship_spmv_csr_set_variant(0)
