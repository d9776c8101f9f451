"""Golden results of the reference for BASELINE config 4 at ITS scale: the Queen_4147 stand-in of gen_queen_class.c (4.1 M rows,
2.9e8 non-zeros, scrambled node numbering) written as a symmetric coordinate Matrix Market file and read back by the reference's
own reader (lis_input -> lis_input_mm_csr, src/system/lis_input_mm.c:699-1069, symmetric expansion :986-1036).

Dev container only: oracle/_ref (Lis 2.1.11 from /root/reference/src, 1 OpenMP thread).  Per case ("mini": 5 184 rows, runs in the
CPU suite; "full": Queen's scale, runs in the GPU suite) the fixture keeps
  csr_sha256   sha256 of ptr / index / value as the reference's reader leaves them (the in-row order fixes every later bit)
  y_sha256     sha256 of y = A x for x_i = cos(0.01 i) + 1.25, lis_matvec of the reference
  solves       iteration count, status, final relative residual and the first residual-history entries of GMRES(30), BiCGSTAB and
               CG + Jacobi with b = A*1 (test/test1.c:138-139, rhs mode 2), x0 = 0, tol 1e-12
  seconds      what the reference's reader took (context for profiles/)
The full case needs ~12 GB of RAM and about ten minutes (two fgets/sscanf passes over a 3.3 GB file, single-thread solves).
    python tests/golden/make_golden_queen_class.py [mini] [full]
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import lisdrv        # noqa: E402
import orc           # noqa: E402
import queen_class   # noqa: E402
from lis_amd import _capi as capi  # noqa: E402

SOLVES = ("-i gmres -restart 30 -p none", "-i bicgstab -p none", "-i cg -p jacobi")
OUT = os.path.join(HERE, "queen_class_golden.json")


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(ref, case):
    path, rows, stored = queen_class.generate(case)
    try:
        A, b, x = capi.PM(), capi.PV(), capi.PV()
        assert ref.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0
        assert ref.lis_vector_create(capi.LIS_COMM_WORLD, C.byref(b)) == 0 and ref.lis_vector_create(capi.LIS_COMM_WORLD, C.byref(x)) == 0
        t0 = time.time()
        assert ref.lis_input(A, b, x, path.encode()) == 0
        secs = time.time() - t0
        size = os.path.getsize(path)
    finally:
        os.unlink(path)
    a = A.contents
    n, nnz = a.n, a.nnz
    assert n == rows and nnz == 2 * stored - n
    view = lambda p, cnt: np.ctypeslib.as_array(p, shape=(cnt,))      # noqa: E731
    case_out = {"G": queen_class.CASES[case][0], "band": queen_class.CASES[case][1], "n": n, "stored_entries": stored, "nnz": nnz,
                "file_bytes": size, "reference_reader_seconds": round(secs, 2),
                "csr_sha256": sha(view(a.ptr, n + 1), view(a.index, nnz), view(a.value, nnz)),
                "max_row": int(np.diff(view(a.ptr, n + 1)).max()), "solves": {}}
    xs = np.cos(np.arange(n) * 0.01) + 1.25
    y = lisdrv.matvec(ref, A, xs)
    case_out["y_sha256"] = sha(y)
    rhs = lisdrv.matvec(ref, A, np.ones(n))
    for opts in SOLVES:
        res = lisdrv.solve(ref, A, rhs, opts + " -tol 1e-12 -maxiter 2000 -print mem")
        case_out["solves"][opts] = {"iter": int(res["iter"]), "status": int(res["status"]), "resid": float(res["resid"]),
                                    "rhistory_head": [float(v) for v in res["rhistory"][:6]], "itime": round(float(res["itime"]), 3)}
        print(case, opts, res["iter"], res["status"], res["resid"], flush=True)
    ref.lis_matrix_destroy(A)
    return case_out


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out["_source"] = ("Lis 2.1.11 compiled from /root/reference by oracle/Makefile, 1 OpenMP thread: lis_input of the file written by "
                      "tests/golden/gen_queen_class.c, y = A*(cos(0.01 i) + 1.25), solves with b = A*1, x0 = 0, tol 1e-12 "
                      "(tests/golden/make_golden_queen_class.py)")
    for case in (sys.argv[1:] or ["mini", "full"]):
        out[case] = run(ref, case)
        json.dump(out, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
