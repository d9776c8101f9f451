"""Golden vectors of the reference for what round 4 added on split matrices (A = L + D + U):

  y_t        y = A^T x through lis_matvech on the SPLIT matrix, CSR (src/matvec/lis_matvec_csr.c:125-160, the OpenMP branch at 1 thread:
             off-diagonal scatter, then D.*x + w) and square-block BSR (lis_matvec_bsr.c:878-925: one chain per entry, D blocks first)
  scale      lis_matrix_scale(A, b, d, action) on the split matrix, jacobi (1) and symm_diag (2): L / U / D values, b, d as the reference
             leaves them (lis_matrix_csr.c:617-632, :661-676; lis_matrix_bsr.c:820-855, :895-935), and y = lis_matvec(A, x) afterwards
  bicg       BiCG with `-scale jacobi -storage bsr` (the solve that needs A^T x of a split matrix): count, status, x

Dev container only: oracle/_ref (Lis 2.1.11 from /root/reference/src), 1 OpenMP thread.  Inputs are those of make_golden_split.py.
    python tests/golden/make_golden_split_t.py
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
from make_golden_split import matrices  # noqa: E402
from make_golden_bscale import matrix as bscale_matrix  # noqa: E402

T_CASES = [("csr", 0), ("bsr", 1), ("bsr", 2), ("bsr", 3), ("bsr", 4), ("bsr", 5)]
S_CASES = [("csr", 0), ("bsr", 2), ("bsr", 3)]
BICG = [("p3d_7x6x5", "-i bicg", 2), ("nonsym_61", "-i bicg", 2), ("nonsym_61", "-i bicg", 1), ("p3d_odd_5x5x3", "-i bicg", 2),
        ("nonsym_61", "-i bicr", 2)]


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    for name, (ptr, idx, val) in matrices():
        n = len(ptr) - 1
        x = np.sin(np.arange(n) * 0.7) + 0.25
        x[::5] = -0.0
        for fmt, bs in T_CASES:
            A = lisdrv.make_csr(ref, ptr, idx, val)
            B = A if fmt == "csr" else lisdrv.convert(ref, A, fmt, bs, bs)
            key = f"{name}/{fmt}{bs if bs else ''}"
            out[key + "/y_t_unsplit"] = lisdrv.matvech(ref, B, x)
            assert ref.lis_matrix_split(B) == 0
            out[key + "/y_t"] = lisdrv.matvech(ref, B, x)
            print(key, "split A^T x differs from the unsplit one in", int((out[key + "/y_t"] != out[key + "/y_t_unsplit"]).sum()), "of", n, "rows")
        for fmt, bs in S_CASES:
            for action in (1, 2):
                A = lisdrv.make_csr(ref, ptr, idx, val)
                B = A if fmt == "csr" else lisdrv.convert(ref, A, fmt, bs, bs)
                assert ref.lis_matrix_split(B) == 0
                b0 = np.cos(np.arange(n) * 0.3) + 2.0
                vb, vd = lisdrv.new_vector(ref, B, b0), lisdrv.new_vector(ref, B)
                assert ref.lis_matrix_scale(B, vb, vd, action) == 0
                key = f"{name}/{fmt}{bs if bs else ''}/scale{action}"
                parts = lisdrv.split_arrays(B)
                out[key + "/L"], out[key + "/U"], out[key + "/D"] = parts["L"]["value"], parts["U"]["value"], parts["D"]
                out[key + "/b"], out[key + "/d"] = lisdrv.get_vector(ref, vb, n), lisdrv.get_vector(ref, vd, n)
                out[key + "/y"] = lisdrv.matvec(ref, B, x)
    for name, opts, block in BICG:
        ptr, idx, val = bscale_matrix(name)
        n = len(ptr) - 1
        b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.3) + 2.0)
        A = lisdrv.make_csr(ref, ptr, idx, val)
        vb, vx = lisdrv.new_vector(ref, A, b), lisdrv.new_vector(ref, A)
        S = capi.PS()
        assert ref.lis_solver_create(C.byref(S)) == 0
        full = f"{opts} -scale jacobi -storage bsr -storage_block {block} -tol 1e-12 -maxiter 500 -print mem"
        assert ref.lis_solver_set_option(full.encode(), S) == 0
        assert ref.lis_solve(A, vb, vx, S) == 0
        key = f"bicg/{name}/{opts.replace(' ', '_')}/b{block}"
        it = S.contents.iter
        out[key + "/opts"] = np.frombuffer(full.encode(), np.uint8)
        out[key + "/iter_status"] = np.array([it, S.contents.retcode])
        out[key + "/x"] = lisdrv.get_vector(ref, vx, n)
        out[key + "/rhistory"] = np.ctypeslib.as_array(S.contents.rhistory, shape=(min(it, 500) + 1,)).copy()
        print(key, it, S.contents.retcode, S.contents.resid)
        ref.lis_solver_destroy(S)
    np.savez_compressed(os.path.join(HERE, "split_t_golden.npz"), **out)


if __name__ == "__main__":
    main()
