"""Known answers at sizes too large for bit fixtures: iteration counts of the reference itself
(oracle/_ref/liblis_ref.so, 8 OpenMP threads; CG counts do not depend on the thread count).

    python tests/golden/make_known.py        # writes tests/golden/known_answers.json  (~2 min)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lisdrv  # noqa: E402
import orc     # noqa: E402

ref = lisdrv.open_lib(orc.REF_SO, threads=8)
out = {"_source": "Lis 2.1.11 compiled from /root/reference by oracle/Makefile, 8 OpenMP threads, test3.c matrix, b=A*1, x0=0, tol 1e-12"}
for N in (64, 128, 256):
    ptr, idx, val = orc.poisson3d(N, N, N)
    A = lisdrv.make_csr(ref, ptr, idx, val)
    b = lisdrv.matvec(ref, A, np.ones(N ** 3))
    r = lisdrv.solve(ref, A, b, "-i cg -p jacobi -tol 1e-12 -maxiter 5000")
    out[f"cg_jacobi_{N}"] = {"iter": r["iter"], "resid": r["resid"]}
    if N <= 128:
        r = lisdrv.solve(ref, A, b, "-i bicgstab -p none -tol 1e-12 -maxiter 5000")
        out[f"bicgstab_none_{N}_8thr"] = {"iter": r["iter"], "resid": r["resid"]}
        r = lisdrv.solve(ref, A, b, "-i gmres -restart 30 -p none -tol 1e-12 -maxiter 5000")
        out[f"gmres30_none_{N}_8thr"] = {"iter": r["iter"], "resid": r["resid"]}
    ref.lis_matrix_destroy(A)
    print(N, out, flush=True)
json.dump(out, open(os.path.join(HERE, "known_answers.json"), "w"), indent=1)
