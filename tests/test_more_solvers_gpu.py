"""The other Krylov solvers of Lis (CGS, CR, GPBiCG, TFQMR, BiCGSafe, Orthomin, BiCR, CRS, BiCRSTAB,
GPBiCR, BiCRSafe, FGMRES, MINRES, COCG, COCR, IDR(s), BiCGSTAB(l)) on the GPU against
what the reference itself produced for the same systems (tests/golden/solvers_golden.npz, make_golden_solvers.py).

Element-wise arithmetic is bit-identical; the reductions are trees, so a recurrence can part from the reference's
by an iteration or two on ill-conditioned cases (the reference's own count moves with OMP_NUM_THREADS).
Bars: status equal; iteration count equal on the well-conditioned cases, within max(3, 10 %) otherwise; first
residuals equal to 1e-8 relative; solution within 1e-8 of the reference's; LIS_MAXITER bookkeeping identical.
"""
import os
import sys

import numpy as np
import pytest

import lis_amd
import lisdrv
import orc
from lis_amd import _capi as capi

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden_scale import test_matrix as nonsym_matrix  # noqa: E402  (a generator, no reference needed)

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "solvers_golden.npz"))
CASES = sorted(k.split("/")[0] for k in G.files if k.endswith("/opts"))


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(0)
    return lib


@pytest.mark.parametrize("case", CASES)
def test_solver_matches_reference(lib, case):
    solver, precon, mat = case.split("_")          # "idrs4" = IDR(s) with -irestart 4, "bicgstabl4" = BiCGSTAB(l) with -ell 4
    ptr, idx, val = orc.poisson3d(8, 7, 6) if mat == "p3d" else nonsym_matrix(n=120, seed=9)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.ones(n))
    opts = bytes(G[case + "/opts"]).decode()
    A = lisdrv.make_csr(lib, ptr, idx, val)
    out = lisdrv.solve(lib, A, b, opts)
    it_ref, st_ref = (int(v) for v in G[case + "/iter_status"])
    assert out["err"] == 0 and out["status"] == st_ref
    if st_ref == 0:
        # FGMRES(5) on the Poisson case stops at 9.9976e-13 against a tolerance of 1e-12 in the reference: a 0.02 % margin
        # that any change of reduction order flips by one iteration -- it gets the loose bar
        if mat == "p3d" and solver != "fgmres":
            assert out["iter"] == it_ref, (case, out["iter"], it_ref)
        else:
            # IDR(s) without a preconditioner on the non-symmetric matrix wanders for 100+ iterations before it drops;
            # where it drops moves by 15-20 % with the reduction order (same first residuals, same P): 25 % bar there
            slack = it_ref // 4 if solver.startswith("idr") else it_ref // 10
            assert abs(out["iter"] - it_ref) <= max(3, slack), (case, out["iter"], it_ref)
        assert out["resid"] <= 1e-12
        assert np.allclose(out["x"], G[case + "/x"], rtol=0, atol=1e-8)
    else:
        assert out["iter"] == it_ref                        # LIS_MAXITER: maxiter + 1
    k = min(out["iter"], it_ref, 5)
    assert np.allclose(out["rhistory"][1:k + 1], G[case + "/rhistory"][1:k + 1], rtol=1e-8, atol=0), case
    cut = lisdrv.solve(lib, A, b, opts.replace("-maxiter 400", "-maxiter 3"))
    assert [cut["iter"], cut["status"]] == [int(v) for v in G[case + "/cut_iter_status"]]       # 4 (IDR(1): 5), LIS_MAXITER
    assert cut["status"] == capi.LIS_MAXITER
    assert np.allclose(cut["x"], G[case + "/cut_x"], rtol=1e-9, atol=1e-12)


def test_unserved_solver_says_so(lib):
    ptr, idx, val = orc.poisson1d(10)
    A = lisdrv.make_csr(lib, ptr, idx, val)
    out = lisdrv.solve(lib, A, np.ones(10), "-i sor")                  # Gauss-Seidel / SOR: not served, said loudly
    assert out["err"] == capi.LIS_ERR_NOT_IMPLEMENTED


@pytest.mark.parametrize("case", CASES)
def test_solver_is_the_reference_bit_for_bit_in_the_ordered_mode(lib, case):
    """With lis_amd_set_reference_reductions(1) the sums are the reference's one-thread sums, and lis_solver_more.c restates the reference's recurrences
    statement by statement: count, status, the whole residual history and the solution must be the reference's in every bit (the fixtures were made at one
    OpenMP thread).  No slack here, whatever the conditioning."""
    solver, precon, mat = case.split("_")
    ptr, idx, val = orc.poisson3d(8, 7, 6) if mat == "p3d" else nonsym_matrix(n=120, seed=9)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.ones(n))
    opts = bytes(G[case + "/opts"]).decode()
    A = lisdrv.make_csr(lib, ptr, idx, val)
    assert lib.dll.lis_amd_set_reference_reductions(1) == 0
    try:
        out = lisdrv.solve(lib, A, b, opts)
    finally:
        lib.dll.lis_amd_set_reference_reductions(0)
    it_ref, st_ref = (int(v) for v in G[case + "/iter_status"])
    assert (out["iter"], out["status"]) == (it_ref, st_ref), (case, out["iter"], out["status"], it_ref, st_ref)
    want = G[case + "/rhistory"]
    assert len(out["rhistory"]) == len(want)
    diff = np.flatnonzero(out["rhistory"].view(np.int64) != want.view(np.int64))
    assert diff.size == 0, (case, "first differing history entry", int(diff[0]), out["rhistory"][diff[0]].hex(), want[diff[0]].hex())
    assert np.array_equal(out["x"].view(np.int64), G[case + "/x"].view(np.int64)), case
    lib.lis_matrix_destroy(A)
