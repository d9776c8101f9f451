"""BASELINE config 5: the same 3-D 7-point Poisson matrix in every storage format, through the Lis API.

    python tests/perf/format_sweep.py [N] [--solve]      # cubic grid edge, default 256
Per format: lis_matvec ms (HIP events on the library's stream), GFLOP/s on the TRUE non-zeros, algorithmic GB/s
(SURVEY 8d byte counts) and, with --solve, CG+Jacobi iterations and it/s.  y is checked against the CSR result
(bit equality where the reference's summation order is the same) and ||A*1||_2 against the closed form."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
import lisdrv   # noqa: E402
import orc      # noqa: E402
from lis_amd import _capi as capi, check  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if args else 256
    solve = "--solve" in sys.argv
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(1)
    dll = lib.dll
    dll.lis_amd_stream.restype = C.c_void_p
    n = N ** 3
    t0 = time.time()
    ptr, idx, val = orc.poisson3d(N, N, N, sort_cols=True)
    nnz = len(idx)
    print(f"N={N} n={n} nnz={nnz} (host build {time.time() - t0:.1f}s)", flush=True)
    A = lisdrv.make_csr(lib, ptr, idx, val)
    want = np.sqrt(6 * (N - 2) ** 2 + 12 * (N - 2) * 4 + 8 * 9.0)
    y_csr = None
    bytes_alg = {"csr": 12 * nnz + 20 * n + 4, "csc": 12 * nnz + 20 * n + 4, "ell": 12 * 7 * n + 16 * n, "dia": 8 * 7 * n + 16 * n,
                 "jad": 12 * nnz + 20 * n + 4,      # served from a row-ordered HBM layout: CSR's bytes
                 "bsr": None}
    for fmt in ("csr", "ell", "dia", "jad", "bsr", "csc"):
        t0 = time.time()
        B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt)
        tconv = time.time() - t0
        vx, vy = lisdrv.new_vector(lib, B), lisdrv.new_vector(lib, B)
        assert lib.lis_vector_set_all(1.0, vx) == 0
        for _ in range(5):
            assert lib.lis_matvec(B, vx, vy) == 0
        timer = C.c_void_p()
        check(lib.liship_timer_create(C.byref(timer)))
        stream = dll.lis_amd_stream()
        reps = 30
        check(lib.liship_timer_start(timer, stream))
        for _ in range(reps):
            assert lib.lis_matvec(B, vx, vy) == 0
        check(lib.liship_timer_stop(timer, stream))
        ms = C.c_float()
        check(lib.liship_timer_elapsed_ms(timer, C.byref(ms)))
        ms = ms.value / reps
        y = lisdrv.get_vector(lib, vy, n)
        nrm = float(np.sqrt(np.sum(y * y)))
        if y_csr is None:
            y_csr = y
        same = bool(np.array_equal(y, y_csr))
        b = bytes_alg[fmt]
        if fmt == "bsr":
            b = 8 * B.contents.bnnz * 4 + 4 * B.contents.bnnz + 4 * (B.contents.nr + 1) + 16 * n
        line = (f"{fmt}: {ms:.4f} ms  {2 * nnz / ms / 1e6:.1f} GFLOP/s  {b / ms / 1e6:.0f} GB/s alg ({b / ms / 1e6 / 80:.1f}% of 8 TB/s)"
                f"  ||A*1||={nrm:.6e} (want {want:.6e})  y==y_csr:{same}  convert {tconv:.3f}s")
        if solve:
            bb = lisdrv.new_vector(lib, B)
            assert lib.lis_matvec(B, vx, bb) == 0
            xs = lisdrv.new_vector(lib, B)
            S = capi.PS()
            lib.lis_solver_create(C.byref(S))
            lib.lis_solver_set_option(b"-i cg -p jacobi -tol 1e-12 -maxiter 2000", S)
            assert lib.lis_solve(B, bb, xs, S) == 0
            line += f"  CG+Jacobi: {S.contents.iter} it, {S.contents.iter / S.contents.itime:.1f} it/s, resid {S.contents.resid:.2e}"
            lib.lis_solver_destroy(S)
            lib.lis_vector_destroy(bb); lib.lis_vector_destroy(xs)
        print(line, flush=True)
        lib.lis_vector_destroy(vx); lib.lis_vector_destroy(vy)
        if B is not A:
            lib.lis_matrix_destroy(B)


if __name__ == "__main__":
    main()
