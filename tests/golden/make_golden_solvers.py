"""Golden results of the reference for the solvers of lis_solver_more.c (CGS, CR, GPBiCG, TFQMR, BiCGSafe, Orthomin, BiCR, CRS,
BiCRSTAB, GPBiCR, BiCRSafe, FGMRES, MINRES, COCG, COCR, IDR(s), BiCGSTAB(l)).

Dev container only: oracle/_ref (Lis 2.1.11 from /root/reference/src, 1 OpenMP thread) through lis_solve; stores the
iteration count, status, residual history and solution per case.
    python tests/golden/make_golden_solvers.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import lisdrv  # noqa: E402
import orc     # noqa: E402

sys.path.insert(0, HERE)
from make_golden_scale import test_matrix  # noqa: E402

CASES = []
for solver in ("cgs", "cr", "gpbicg", "tfqmr", "bicgsafe", "orthomin",
               "bicr", "crs", "bicrstab", "gpbicr", "bicrsafe", "fgmres", "minres", "cocg", "cocr", "idrs", "idrs4", "bicgstabl", "bicgstabl4", "idr1", "jacobi"):
    for precon in ("none", "jacobi"):
        for mat in ("p3d", "nonsym"):
            if solver in ("cr", "minres", "cocg", "cocr") and mat == "nonsym":
                continue                                   # symmetric systems only
            if solver == "jacobi" and precon != "none":
                continue                                   # the stationary solver is served without a preconditioner
            CASES.append((solver, precon, mat))


def matrix(name):
    return orc.poisson3d(8, 7, 6) if name == "p3d" else test_matrix(n=120, seed=9)


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    for solver, precon, mat in CASES:
        ptr, idx, val = matrix(mat)
        n = len(ptr) - 1
        b = orc.spmv_csr(ptr, idx, val, np.ones(n))
        A = lisdrv.make_csr(ref, ptr, idx, val)
        opts = f"-i {solver.rstrip('4')} -p {precon} -tol 1e-12 -maxiter 400 -print mem" + (" -restart 5" if solver in ("orthomin", "fgmres") else "")
        if solver == "idrs4":
            opts += " -irestart 4"
        if solver == "bicgstabl4":
            opts += " -ell 4"
        res = lisdrv.solve(ref, A, b, opts)
        key = f"{solver}_{precon}_{mat}"
        out[key + "/iter_status"] = np.array([res["iter"], res["status"]])
        out[key + "/x"], out[key + "/rhistory"] = res["x"], res["rhistory"]
        out[key + "/opts"] = np.frombuffer(opts.encode(), np.uint8)
        # a run cut after 3 iterations: LIS_MAXITER bookkeeping
        res3 = lisdrv.solve(ref, A, b, opts.replace("-maxiter 400", "-maxiter 3"))
        out[key + "/cut_iter_status"] = np.array([res3["iter"], res3["status"]])
        out[key + "/cut_x"] = res3["x"]
        print(key, res["iter"], res["status"], res["resid"], "| cut:", res3["iter"], res3["status"])
    np.savez_compressed(os.path.join(HERE, "solvers_golden.npz"), **out)


if __name__ == "__main__":
    main()
