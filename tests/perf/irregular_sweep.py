"""BASELINE config 4 stand-in: irregular CSR, GMRES(30), one GPU.

SuiteSparse Queen_4147 cannot be fetched here (no network), so two synthetic matrices take its place:
  fem3   3 unknowns per node of a G^3 grid, 27-node connectivity (3x3 blocks, up to 81 non-zeros per row, boundary
         rows shorter) -- the sparsity class of Queen_4147 (3-D structural FEM); symmetric, diagonally dominant
  zipf   row lengths drawn from a heavy-tailed law (1 .. 200 000 per row), random columns: the load-balance stress
         for the merge-path row split
  mesh   (IRREG_MESH_NODES=N, default 1 000 000) an unstructured 3-D mesh, one unknown per node, ragged rows of 8 .. 40 entries (mean ~19), varying
         coefficients, numbered along a coarse Morton curve and at random inside a cell: the class none of the plan's special forms catch (orc.unstructured_mesh)
    python tests/perf/irregular_sweep.py [G] [--gmres-iters K]
Prints SpMV ms / GFLOP/s / algorithmic GB/s (12*nnz + 20*n bytes) with y checked bit for bit against the CPU oracle,
and GMRES(30) iterations per second through lis_solve."""
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


fem3 = orc.fem3


def zipf(n, seed=3):
    rng = np.random.default_rng(seed)
    lens = np.minimum((rng.pareto(1.3, n) * 8 + 1).astype(np.int64), 200000)
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[-1])
    idx = rng.integers(0, n, nnz, dtype=np.int32)
    val = rng.uniform(-1, 1, nnz)
    return ptr.astype(np.int32), idx, val, n


def time_spmv(lib, dll, A, vx, vy, reps=30):
    for _ in range(5):
        assert lib.lis_matvec(A, vx, vy) == 0
    timer = C.c_void_p()
    check(lib.liship_timer_create(C.byref(timer)))
    stream = dll.lis_amd_stream()
    check(lib.liship_timer_start(timer, stream))
    for _ in range(reps):
        assert lib.lis_matvec(A, vx, vy) == 0
    check(lib.liship_timer_stop(timer, stream))
    ms = C.c_float()
    check(lib.liship_timer_elapsed_ms(timer, C.byref(ms)))
    return ms.value / reps


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    G = int(args[0]) if args else 80
    iters = int(sys.argv[sys.argv.index("--gmres-iters") + 1]) if "--gmres-iters" in sys.argv else 150
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(1)
    dll = lib.dll
    dll.lis_amd_stream.restype = C.c_void_p
    variants = [int(v, 0) for v in os.environ.get("SWEEP_VARIANTS", "0").split(",")]
    if os.environ.get("IRREG_UNIFORM_ROWS"):           # A/B of the uniform-length row sums (0: skewed sums everywhere)
        check(lib.liship_spmv_csr_set_uniform_rows(int(os.environ["IRREG_UNIFORM_ROWS"])))
    if os.environ.get("IRREG_ROUND3") == "1":          # A/B: the round-3 form of the block-local kernel (4096-item blocks, positions through LDS)
        check(lib.liship_spmv_csr_set_local_register_positions(0))
    only = os.environ.get("IRREG_ONLY")                # "fem3" / "zipf": one of the two (profiling runs)
    def mesh():
        kmin, kmax = (int(t) for t in os.environ.get("IRREG_MESH_K", "6,22").split(","))      # neighbours a node asks for: the mean row is ~ 1 + 1.3 (kmin + kmax) / 2
        p, i, v = orc.unstructured_mesh(int(os.environ.get("IRREG_MESH_NODES", "1000000")), kmin=kmin, kmax=kmax)
        return p, i, v, len(p) - 1
    for name, gen in (("fem3", lambda: fem3(G)[:4]), ("zipf", lambda: zipf(2_000_000)), ("mesh", mesh)):
        if only and name != only:
            continue
        t0 = time.time()
        ptr, idx, val, n = gen()
        nnz = len(idx)
        lens = np.diff(ptr)
        print(f"{name}: n={n} nnz={nnz} rows: min {lens.min()} mean {lens.mean():.1f} max {lens.max()} (built in {time.time() - t0:.1f}s)", flush=True)
        x = np.cos(np.arange(n) * 0.01) + 1.25
        yref = orc.spmv_csr(ptr, idx, val, x)
        b = 12 * nnz + 20 * n + 4
        for variant in variants:                       # development knob of the CSR kernel (spmv_csr.hip); 0 = shipped
            lib.liship_spmv_csr_set_variant(variant)
            A = lisdrv.make_csr(lib, ptr, idx, val)
            vx, vy = lisdrv.new_vector(lib, A, x), lisdrv.new_vector(lib, A)
            ms = time_spmv(lib, dll, A, vx, vy)
            exact = bool(np.array_equal(lisdrv.get_vector(lib, vy, n), yref))
            print(f"{name}: variant {variant:#x}: SpMV {ms:.4f} ms  {2 * nnz / ms / 1e6:.1f} GFLOP/s  {b / ms / 1e6:.0f} GB/s alg "
                  f"({b / ms / 1e6 / 80:.1f}% of 8 TB/s)  bit-exact vs oracle: {exact}", flush=True)
            if variant != variants[-1]:
                lib.lis_matrix_destroy(A)
        lib.liship_spmv_csr_set_variant(0)
        if name in ("fem3", "mesh") and iters > 0:
            bb = lisdrv.new_vector(lib, A)
            assert lib.lis_matvec(A, vx, bb) == 0           # b = A * x_true, x_true = cos(0.01 i) + 1.25
            for opts in ("-i gmres -restart 30 -p none", "-i gmres -restart 30 -p jacobi", "-i bicgstab -p none", "-i cg -p jacobi"):
                xs = lisdrv.new_vector(lib, A)
                S = capi.PS()
                lib.lis_solver_create(C.byref(S))
                lib.lis_solver_set_option((opts + f" -tol 1e-12 -maxiter {iters}").encode(), S)
                assert lib.lis_solve(A, bb, xs, S) == 0
                it = min(S.contents.iter, iters)
                print(f"{name}: {opts}: {S.contents.iter} it (status {S.contents.retcode}), {it / S.contents.itime:.1f} it/s, "
                      f"rel. residual {S.contents.resid:.3e}", flush=True)
                lib.lis_solver_destroy(S)
                lib.lis_vector_destroy(xs)
        lib.lis_matrix_destroy(A)


if __name__ == "__main__":
    main()
