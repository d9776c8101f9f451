"""Golden vectors for `-scale` (lis_matrix_scale, src/matrix/lis_matrix_ops.c:579) from the reference itself.

Dev container only.  For one non-symmetric matrix with a full diagonal, in every storage format, both actions
(1 = jacobi, 2 = symm_diag): the scaled value array, b and d exactly as the reference leaves them; and lis_solve
with `-scale` for the four served solvers: iteration count, status, x.
    python tests/golden/make_golden_scale.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import lisdrv  # noqa: E402
import orc     # noqa: E402
from lis_amd import _capi as capi  # noqa: E402


def test_matrix(n=61, seed=5):
    """random pattern, every row has a positive or negative diagonal of size ~ row sum (stored wherever it falls)"""
    ptr, idx, val = orc.random_csr(n, 5, seed=seed, empty_rows=False)
    rng = np.random.default_rng(seed)
    nptr = (ptr + np.arange(n + 1)).astype(np.int32)
    nidx, nval = np.empty(len(idx) + n, np.int32), np.empty(len(idx) + n)
    for r in range(n):
        s, e = ptr[r], ptr[r + 1]
        cols = np.where(idx[s:e] == r, (r + 1) % n, idx[s:e])
        k = int(rng.integers(0, e - s + 1))                       # position of the diagonal inside the row
        nidx[nptr[r]:nptr[r + 1]] = np.insert(cols, k, r)
        nval[nptr[r]:nptr[r + 1]] = np.insert(val[s:e], k, (np.abs(val[s:e]).sum() + 1.0) * (1 if r % 3 else -1))
    return nptr, nidx, nval


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    ptr, idx, val = test_matrix()
    n = len(ptr) - 1
    b = np.cos(np.arange(n) * 0.3) + 2.0
    out["ptr"], out["idx"], out["val"], out["b"] = ptr, idx, val, b
    for action in (1, 2):
        for fmt in ("csr", "csc", "ell", "dia", "jad", "bsr"):
            A = lisdrv.make_csr(ref, ptr, idx, val)
            B = A if fmt == "csr" else lisdrv.convert(ref, A, fmt)
            vb, vd = lisdrv.new_vector(ref, B, b), lisdrv.new_vector(ref, B)
            assert ref.lis_matrix_scale(B, vb, vd, action) == 0
            arrs = lisdrv.matrix_arrays(B)
            out[f"scale{action}/{fmt}/value"] = arrs["value"]
            out[f"scale{action}/{fmt}/b"] = lisdrv.get_vector(ref, vb, n)
            out[f"scale{action}/{fmt}/d"] = lisdrv.get_vector(ref, vd, n)
            x0 = np.sin(np.arange(n) * 0.7) + 1.5
            out[f"scale{action}/{fmt}/y"] = lisdrv.matvec(ref, B, x0)
    # solves: symmetric Poisson for CG, the non-symmetric matrix for the others
    p3 = orc.poisson3d(7, 6, 5)
    for name, mat, opts in (("cg_jacobi", p3, "-i cg -scale jacobi"), ("cg_symm_pjac", p3, "-i cg -p jacobi -scale symm_diag"),
                            ("bicgstab_jacobi", (ptr, idx, val), "-i bicgstab -scale jacobi"),
                            ("gmres_symm", (ptr, idx, val), "-i gmres -restart 20 -scale symm_diag"),
                            ("bicg_jacobi_pjac", (ptr, idx, val), "-i bicg -p jacobi -scale jacobi"),
                            ("cg_jacobi_ell", p3, "-i cg -scale jacobi -storage ell")):
        mp, mi, mv = mat
        nn = len(mp) - 1
        bb = orc.spmv_csr(mp, mi, mv, np.ones(nn))
        A = lisdrv.make_csr(ref, mp, mi, mv)
        res = lisdrv.solve(ref, A, bb, opts + " -tol 1e-12 -maxiter 500 -print mem")
        out[f"solve/{name}/iter_status"] = np.array([res["iter"], res["status"]])
        out[f"solve/{name}/x"], out[f"solve/{name}/rhistory"] = res["x"], res["rhistory"]
        out[f"solve/{name}/opts"] = np.frombuffer(opts.encode(), np.uint8)
        out[f"solve/{name}/grid"] = np.array([7, 6, 5] if mat is p3 else [0, 0, 0])
        out[f"solve/{name}/A_value_after"] = lisdrv.matrix_arrays(A)["value"]      # A stays scaled (is_scaled)
        print(name, res["iter"], res["status"], res["resid"])
    np.savez_compressed(os.path.join(HERE, "scale_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
