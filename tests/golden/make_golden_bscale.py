"""Golden results of the reference for `-scale jacobi -storage bsr` (block-diagonal scaling, src/solver/lis_solver.c:659-690:
lis_matrix_split, lis_matrix_diag_inverse, lis_matrix_bscale_bsr, lis_matrix_diag_matvec) with 1 x 1 and 2 x 2 blocks -- the
block sizes the reference scales completely (its 3 x 3 case writes back 8 of 9 entries, larger blocks are not scaled at all).

Dev container only: oracle/_ref, 1 OpenMP thread.  Stored per case: L / U / D values and b as the reference leaves them after
lis_solve, iteration count, status, x.
    python tests/golden/make_golden_bscale.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import lisdrv  # noqa: E402
import orc     # noqa: E402
from lis_amd import _capi as capi  # noqa: E402
from make_golden_scale import test_matrix  # noqa: E402

CASES = [("p3d_7x6x5", "-i cg", 2), ("p3d_7x6x5", "-i cg -p jacobi", 1), ("nonsym_61", "-i bicgstab", 2), ("nonsym_61", "-i gmres -restart 20", 2),
         ("nonsym_61", "-i bicgstab", 1), ("p3d_odd_5x5x3", "-i cg", 2)]          # 75 rows: the last 2 x 2 block is padded


def matrix(name):
    if name == "p3d_7x6x5":
        return orc.poisson3d(7, 6, 5)
    if name == "p3d_odd_5x5x3":
        return orc.poisson3d(5, 5, 3)
    return test_matrix(61, 5)


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    for name, opts, block in CASES:
        ptr, idx, val = matrix(name)
        n = len(ptr) - 1
        b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.3) + 2.0)
        A = lisdrv.make_csr(ref, ptr, idx, val)
        vb, vx = lisdrv.new_vector(ref, A, b), lisdrv.new_vector(ref, A)
        S = capi.PS()
        assert ref.lis_solver_create(C.byref(S)) == 0
        full = f"{opts} -scale jacobi -storage bsr -storage_block {block} -tol 1e-12 -maxiter 500"
        assert ref.lis_solver_set_option(full.encode(), S) == 0
        assert ref.lis_solve(A, vb, vx, S) == 0
        key = f"{name}/{opts.replace(' ', '_')}/b{block}"
        out[key + "/opts"] = np.frombuffer(full.encode(), np.uint8)
        out[key + "/iter_status"] = np.array([S.contents.iter, S.contents.retcode])
        out[key + "/x"] = lisdrv.get_vector(ref, vx, n)
        out[key + "/b_scaled"] = lisdrv.get_vector(ref, vb, n)
        assert A.contents.matrix_type == capi.LIS_MATRIX_BSR and A.contents.is_splited
        parts = lisdrv.split_arrays(A)
        out[key + "/L"], out[key + "/U"], out[key + "/D"] = parts["L"]["value"], parts["U"]["value"], parts["D"]
        print(key, S.contents.iter, S.contents.retcode, S.contents.resid)
        ref.lis_solver_destroy(S)
    np.savez_compressed(os.path.join(HERE, "bscale_golden.npz"), **out)


if __name__ == "__main__":
    main()
