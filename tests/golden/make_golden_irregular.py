"""Golden results of the reference for BASELINE config 4's class: irregular CSR with long rows, GMRES(30) and friends.

SuiteSparse Queen_4147 itself is neither in the reference tree nor fetchable here; two generated matrices stand in
(tests/orc.py, deterministic numpy code, fingerprinted below so that a drift of the generators is noticed):
  fem3_22   orc.fem3(22): 3 unknowns per node of a 22^3 grid, 27-node connectivity -- 31 944 rows, 2.4 M non-zeros,
            up to 81 per row, symmetric, diagonally dominant (the block-local-columns kernel serves it)
  tail      orc.heavy_tail(30000): Pareto row lengths 1 .. 9000 (mean ~25, rows longer than the LDS stage), random
            columns, non-symmetric, diagonally dominant (the products kernel and its long-row passes serve it)
  mesh_60k  orc.unstructured_mesh(60000) (round 6): an unstructured 3-D mesh, one unknown per node, ragged rows of 7 .. 30 entries,
            varying coefficients, symmetric positive definite -- no row patterns, no blocks, no constant values: the class none of the
            plan's special forms catch (the row-gather kernel on the reference's own arrays serves it)
Dev container only: oracle/_ref (Lis 2.1.11 from /root/reference/src, 1 OpenMP thread) through lis_matvec / lis_solve, with
b = A * x_true as test/test1.c builds it in rhs mode 2 (test1.c:138-139), but with x_true = cos(0.01 i) + 1.25 instead of 1: the
rows of fem3 sum to 1, so b = A*1 = 1 would be solved in one iteration.  Stored per case: iteration count (at 1 thread, and at 2 / 4 / 8 / 16 threads -- the spread the reference's own chunked sums give),
status, final relative residual, the first residual-history entries, and a checksum of y = A*x.
    python tests/golden/make_golden_irregular.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import lisdrv  # noqa: E402
import orc     # noqa: E402

SOLVES = {
    "fem3_22": ("-i gmres -restart 30 -p none", "-i gmres -restart 30 -p jacobi", "-i bicgstab -p none", "-i cg -p jacobi", "-i bicg -p none"),
    "tail": ("-i gmres -restart 30 -p none", "-i gmres -restart 30 -p jacobi", "-i bicgstab -p jacobi", "-i bicg -p jacobi"),
    "mesh_60k": ("-i gmres -restart 30 -p jacobi", "-i bicgstab -p jacobi", "-i cg -p jacobi", "-i bicg -p jacobi"),
}


def matrices():
    ptr, idx, val, _ = orc.fem3(22)
    yield "fem3_22", ptr, idx, val
    yield ("tail",) + orc.heavy_tail(30000)
    yield ("mesh_60k",) + orc.unstructured_mesh(60000)


def fingerprint(ptr, idx, val):
    h = hashlib.sha256()
    for a in (np.ascontiguousarray(ptr, np.int32), np.ascontiguousarray(idx, np.int32), np.ascontiguousarray(val, np.float64)):
        h.update(a.tobytes())
    return h.hexdigest()


def counts_at(threads):
    """iteration counts of every case with the reference's OpenMP team at `threads` (its sums are formed per thread chunk: the counts move with the team size)"""
    ref = lisdrv.open_lib(orc.REF_SO, threads=threads)
    out = {}
    for name, ptr, idx, val in matrices():
        n = len(ptr) - 1
        A = lisdrv.make_csr(ref, ptr, idx, val)
        b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.01) + 1.25)
        out[name] = {}
        for opts in SOLVES[name]:
            res = lisdrv.solve(ref, A, b, opts + " -tol 1e-12 -maxiter 2000 -print none")
            out[name][opts] = [int(res["iter"]), int(res["status"])]
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--threads":
        fd = os.dup(1); os.dup2(2, 1)                                   # (the reference prints its banners on stdout)
        res = counts_at(int(sys.argv[2]))
        os.write(fd, (json.dumps(res) + "\n").encode())
        return
    orc.build()
    import subprocess
    spread = {}
    for t in (2, 4, 8, 16):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--threads", str(t)], capture_output=True, text=True, check=True)
        spread[t] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {"_source": "Lis 2.1.11 compiled from /root/reference by oracle/Makefile, 1 OpenMP thread, b = A*(cos(0.01 i) + 1.25), x0 = 0, tol 1e-12 "
                      "(tests/golden/make_golden_irregular.py)"}
    for name, ptr, idx, val in matrices():
        n = len(ptr) - 1
        A = lisdrv.make_csr(ref, ptr, idx, val)
        x = np.cos(np.arange(n) * 0.01) + 1.25
        vx, vy = lisdrv.new_vector(ref, A, x), lisdrv.new_vector(ref, A)
        assert ref.lis_matvec(A, vx, vy) == 0
        y = lisdrv.get_vector(ref, vy, n)
        assert np.array_equal(y, orc.spmv_csr(ptr, idx, val, x))          # the oracle restates the reference's product
        b = y                                                             # = A * x_true
        lens = np.diff(ptr)
        case = {"n": n, "nnz": int(len(idx)), "max_row": int(lens.max()), "sha256": fingerprint(ptr, idx, val),
                "y_sha256": hashlib.sha256(y.tobytes()).hexdigest(), "solves": {}}
        for opts in SOLVES[name]:
            res = lisdrv.solve(ref, A, b, opts + " -tol 1e-12 -maxiter 2000 -print mem")
            case["solves"][opts] = {"iter": int(res["iter"]), "status": int(res["status"]), "resid": float(res["resid"]),
                                    "rhistory_head": [float(v) for v in res["rhistory"][:6]],
                                    # the reference's own counts with 2 / 4 / 8 / 16 OpenMP threads (all converged): the spread its chunked sums give
                                    "iter_by_threads": dict({"1": int(res["iter"])}, **{str(t): spread[t][name][opts][0] for t in spread})}
            assert all(spread[t][name][opts][1] == res["status"] for t in spread)
            print(name, opts, res["iter"], res["status"], res["resid"])
        out[name] = case
    json.dump(out, open(os.path.join(HERE, "irregular_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
