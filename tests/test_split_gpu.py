"""lis_matvec on split matrices (A = L + D + U) on a real MI355X: the bits the reference's is_splited branches
return (tests/golden/split_golden.npz), for every storage format, and `-scale jacobi -storage bsr` (block scaling)."""
import ctypes as C
import os

import numpy as np
import pytest

import lis_amd
import lisdrv
import orc
from lis_amd import _capi as capi

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_golden.npz"))
MATS = ["p3d_6x5x4", "nonsym_61", "zeros_40"]
CASES = [("csr", 0), ("csc", 0), ("ell", 0), ("dia", 0), ("jad", 0), ("bsr", 1), ("bsr", 2), ("bsr", 3), ("bsr", 4), ("bsr", 5)]


def same_bits(a, b):
    return np.array_equal(np.asarray(a, np.float64).view(np.int64), np.asarray(b, np.float64).view(np.int64))


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(0)
    return lib


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", CASES)
def test_split_product_has_the_reference_bits(lib, name, fmt, bs):
    ptr, idx, val, x = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val", "x"))
    key = f"{name}/{fmt}{bs if bs else ''}"
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt, bs, bs)
    assert same_bits(lisdrv.matvec(lib, B, x), G[key + "/y_unsplit"])
    assert lib.lis_matrix_split(B) == 0
    assert same_bits(lisdrv.matvec(lib, B, x), G[key + "/y_split"])             # D first, then L, then U; signed zeros included
    assert lib.lis_matrix_merge(B) == 0
    # and back: CSR and BSR arrays are REBUILT from the parts in L, D, U order as the reference's merge does (lis_matrix_bsr.c:1337-1395),
    # so the product is the unsplit loop over the merged arrays; the other formats return to the arrays they never lost
    arrs = lisdrv.matrix_arrays(B)
    if fmt == "csr":
        want = orc.spmv_csr(arrs["ptr"], arrs["index"], arrs["value"], x)
    elif fmt == "bsr":
        want = orc.spmv_bsr(arrs["n"], arrs["nr"], bs, bs, arrs["bptr"], arrs["bindex"], arrs["value"], x)
    else:
        want = G[key + "/y_unsplit"]
    assert same_bits(lisdrv.matvec(lib, B, x), want)
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


@pytest.mark.parametrize("fmt", ["csr", "ell", "jad", "bsr"])
def test_solver_on_a_split_matrix(lib, fmt):
    """the Krylov loops run on the split product like on any other (fused epilogues included)"""
    ptr, idx, val = orc.poisson3d(9, 8, 7)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.ones(n))
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt)
    ref = lisdrv.solve(lib, B, b, "-i cg -p jacobi -tol 1e-12 -maxiter 500")
    assert lib.lis_matrix_split(B) == 0
    res = lisdrv.solve(lib, B, b, "-i cg -p jacobi -tol 1e-12 -maxiter 500")
    assert res["status"] == 0 and res["resid"] <= 1e-12 and abs(res["iter"] - ref["iter"]) <= 1
    np.testing.assert_allclose(res["x"], np.ones(n), rtol=0, atol=1e-10)
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


GB = np.load(os.path.join(os.path.dirname(__file__), "golden", "bscale_golden.npz"))
BCASES = sorted({k.rsplit("/", 1)[0] for k in GB.files})


@pytest.mark.parametrize("key", BCASES)
def test_block_scaled_solves_match_the_reference(lib, key):
    """-scale jacobi -storage bsr: A is converted, split and scaled by the inverse diagonal blocks, the iterations run on the split
    product; iteration counts and solutions of the reference (tests/golden/make_golden_bscale.py)"""
    from test_split_cpu import bscale_matrix
    name = key.split("/")[0]
    opts = bytes(GB[key + "/opts"]).decode()
    ptr, idx, val = bscale_matrix(name)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.3) + 2.0)
    A = lisdrv.make_csr(lib, ptr, idx, val)
    vb, vx = lisdrv.new_vector(lib, A, b), lisdrv.new_vector(lib, A)
    S = capi.PS()
    assert lib.lis_solver_create(C.byref(S)) == 0
    assert lib.lis_solver_set_option(opts.encode(), S) == 0
    assert lib.lis_solve(A, vb, vx, S) == 0
    it, st = (int(v) for v in GB[key + "/iter_status"])
    assert S.contents.retcode == st == 0 and S.contents.resid <= 1e-12
    assert abs(S.contents.iter - it) <= (0 if "cg" in opts else 1), (S.contents.iter, it)
    assert A.contents.matrix_type == capi.LIS_MATRIX_BSR and A.contents.is_splited == 1      # converted and split for good, as in the reference
    parts = lisdrv.split_arrays(A)
    assert same_bits(parts["L"]["value"], GB[key + "/L"]) and same_bits(parts["D"], GB[key + "/D"])
    assert same_bits(lisdrv.get_vector(lib, vb, n), GB[key + "/b_scaled"])
    np.testing.assert_allclose(lisdrv.get_vector(lib, vx, n), GB[key + "/x"], rtol=1e-9, atol=1e-11)
    lib.lis_solver_destroy(S)
    lib.lis_matrix_destroy(A)
