"""Generate tests/golden/lis_ref_golden.npz from the reference itself.

Runs ONLY in the dev container: drives oracle/_ref/liblis_ref.so (Lis 2.1.11 compiled from
/root/reference/src by oracle/Makefile) through its public C API at 1 OpenMP thread and stores
inputs' generator parameters + the reference's outputs (bit patterns preserved: float64 arrays).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import lisdrv  # noqa: E402
import orc     # noqa: E402


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    rng = np.random.default_rng(20260928)

    # --- SpMV in every format: 1-D n=100 (spmvtest1), 3-D 6x5x4 (test3 order), 3-D 8^3 sorted (spmvtest3),
    #     and one irregular matrix with empty rows and a long row
    mats = {
        "p1d100": orc.poisson1d(100),
        "p3d_6x5x4": orc.poisson3d(6, 5, 4),
        "p3d_8s": orc.poisson3d(8, 8, 8, sort_cols=True),
        "irr150": orc.random_csr(150, 8, seed=99, long_row=120),
    }
    for name, (ptr, idx, val) in mats.items():
        n = len(ptr) - 1
        x = rng.uniform(-1, 1, n)
        out[f"{name}/ptr"], out[f"{name}/idx"], out[f"{name}/val"], out[f"{name}/x"] = ptr, idx, val, x
        A = lisdrv.make_csr(ref, ptr, idx, val)
        out[f"{name}/y_csr"] = lisdrv.matvec(ref, A, x)
        out[f"{name}/y_ones_nrm2"] = np.array([np.sqrt(np.sum(lisdrv.matvec(ref, A, np.ones(n)) ** 2))])
        for fmt in ("csc", "ell", "dia", "jad", "bsr"):
            A2 = lisdrv.make_csr(ref, ptr, idx, val)       # csr2dia sorts its input in place
            B = lisdrv.convert(ref, A2, fmt)
            out[f"{name}/y_{fmt}"] = lisdrv.matvec(ref, B, x)
            arrs = lisdrv.matrix_arrays(B)
            for k, v in arrs.items():
                if isinstance(v, np.ndarray):
                    out[f"{name}/{fmt}/{k}"] = v
                else:
                    out[f"{name}/{fmt}/{k}"] = np.array([v])
            ref.lis_matrix_destroy(B)
            ref.lis_matrix_destroy(A2)
        ref.lis_matrix_destroy(A)

    # --- Krylov loops, b = A*1, x0 = 0 (test/test3.c:133-160): iteration count, status, residual history, x
    solves = [
        ("cg_jacobi_8", "cg", "jacobi", (8, 8, 8), ""),
        ("cg_none_6x7x5", "cg", "none", (6, 7, 5), ""),
        ("cg_jacobi_16", "cg", "jacobi", (16, 16, 16), ""),
        ("bicgstab_none_8", "bicgstab", "none", (8, 8, 8), ""),
        ("bicgstab_jacobi_12", "bicgstab", "jacobi", (12, 12, 12), ""),
        ("gmres_none_8_r7", "gmres", "none", (8, 8, 8), " -restart 7"),
        ("gmres_jacobi_10_r30", "gmres", "jacobi", (10, 10, 10), " -restart 30"),
    ]
    for name, solver, precon, grid, extra in solves:
        ptr, idx, val = orc.poisson3d(*grid)
        n = len(ptr) - 1
        A = lisdrv.make_csr(ref, ptr, idx, val)
        b = lisdrv.matvec(ref, A, np.ones(n))
        r = lisdrv.solve(ref, A, b, f"-i {solver} -p {precon} -tol 1e-12 -maxiter 1000 -print mem" + extra)
        out[f"solve/{name}/grid"] = np.array(grid)
        out[f"solve/{name}/b"] = b
        out[f"solve/{name}/x"] = r["x"]
        out[f"solve/{name}/iter_status"] = np.array([r["iter"], r["status"]])
        out[f"solve/{name}/resid"] = np.array([r["resid"]])
        out[f"solve/{name}/rhistory"] = r["rhistory"]
        ref.lis_matrix_destroy(A)

    # --- known answers the reference's drivers print
    #     test3 N N N 1 -i cg -p jacobi : 103 iterations at N=32 ; spmvtest3 ||A*1||_2
    ptr, idx, val = orc.poisson3d(32, 32, 32)
    A = lisdrv.make_csr(ref, ptr, idx, val)
    b = lisdrv.matvec(ref, A, np.ones(32 ** 3))
    r = lisdrv.solve(ref, A, b, "-i cg -p jacobi -tol 1e-12 -maxiter 1000")
    out["known/cg_jacobi_32/iter_resid"] = np.array([r["iter"], r["resid"]])
    out["known/p3d_32/nrm2_A1"] = np.array([np.sqrt(np.dot(b, b))])
    r = lisdrv.solve(ref, A, b, "-i bicgstab -p none -tol 1e-12 -maxiter 1000")
    out["known/bicgstab_none_32/iter_resid"] = np.array([r["iter"], r["resid"]])
    r = lisdrv.solve(ref, A, b, "-i gmres -restart 30 -p none -tol 1e-12 -maxiter 1000")
    out["known/gmres30_none_32/iter_resid"] = np.array([r["iter"], r["resid"]])

    path = os.path.join(HERE, "lis_ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
