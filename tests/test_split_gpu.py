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


# ---------------------------------------------------------------- round 4: A^T x of split matrices, scaling of split and HBM-only matrices
GT = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_t_golden.npz"))
T_CASES = [("csr", 0), ("bsr", 1), ("bsr", 2), ("bsr", 3), ("bsr", 4), ("bsr", 5)]
S_CASES = [("csr", 0), ("bsr", 2), ("bsr", 3)]


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", T_CASES)
def test_split_transposed_product_has_the_reference_bits(lib, name, fmt, bs):
    """lis_matvech on a split matrix: CSR = D.*x + (off-diagonal scatter sums) (lis_matvec_csr.c:125-160), BSR = one chain per entry with the
    diagonal blocks' terms first (lis_matvec_bsr.c:878-925); tests/golden/make_golden_split_t.py"""
    ptr, idx, val, x = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val", "x"))
    key = f"{name}/{fmt}{bs if bs else ''}"
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt, bs, bs)
    assert same_bits(lisdrv.matvech(lib, B, x), GT[key + "/y_t_unsplit"])
    assert lib.lis_matrix_split(B) == 0
    assert same_bits(lisdrv.matvech(lib, B, x), GT[key + "/y_t"])
    assert lib.lis_matrix_merge(B) == 0                                   # and the transposed copy follows the matrix back:
    if fmt != "csr" and fmt != "bsr":                                      # (CSR / BSR arrays are rebuilt in L, D, U order by the merge: another walk)
        assert same_bits(lisdrv.matvech(lib, B, x), GT[key + "/y_t_unsplit"])
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", S_CASES)
@pytest.mark.parametrize("action", [1, 2])
def test_scaling_a_split_matrix_leaves_the_reference_arrays(lib, name, fmt, bs, action):
    ptr, idx, val, x = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val", "x"))
    n = len(ptr) - 1
    key = f"{name}/{fmt}{bs if bs else ''}/scale{action}"
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt, bs, bs)
    assert lib.lis_matrix_split(B) == 0
    vb, vd = lisdrv.new_vector(lib, B, np.cos(np.arange(n) * 0.3) + 2.0), lisdrv.new_vector(lib, B)
    assert lib.lis_matrix_scale(B, vb, vd, action) == 0
    parts = lisdrv.split_arrays(B)
    assert same_bits(parts["L"]["value"], GT[key + "/L"]) and same_bits(parts["U"]["value"], GT[key + "/U"]) and same_bits(parts["D"], GT[key + "/D"])
    assert same_bits(lisdrv.get_vector(lib, vb, n), GT[key + "/b"]) and same_bits(lisdrv.get_vector(lib, vd, n), GT[key + "/d"])
    assert same_bits(lisdrv.matvec(lib, B, x), GT[key + "/y"])             # the HBM copy was rebuilt from the scaled parts
    assert B.contents.is_scaled == 1
    lib.lis_vector_destroy(vb); lib.lis_vector_destroy(vd)
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


BICG_CASES = sorted({k.rsplit("/", 1)[0] for k in GT.files if k.startswith("bicg/")})


@pytest.mark.parametrize("key", BICG_CASES)
def test_bicg_type_solvers_on_block_scaled_bsr(lib, key):
    """`-i bicg -scale jacobi -storage bsr`: A is retyped, split and block-scaled, and the dual recurrence multiplies by the TRANSPOSE of the
    split matrix (refused until round 4).  In the reference-order mode the solve is the reference's in every bit: count, residual history, x."""
    from test_split_cpu import bscale_matrix
    name = key.split("/")[1]
    opts = bytes(GT[key + "/opts"]).decode()
    ptr, idx, val = bscale_matrix(name)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.3) + 2.0)
    it, st = (int(v) for v in GT[key + "/iter_status"])
    for ordered in (0, 1):
        A = lisdrv.make_csr(lib, ptr, idx, val)
        lib.dll.lis_amd_set_reference_reductions(1 if ordered else 0)
        try:
            res = lisdrv.solve(lib, A, b, opts)
        finally:
            lib.dll.lis_amd_set_reference_reductions(0)
        assert res["err"] == 0 and res["status"] == st == 0 and res["resid"] <= 1e-12
        if ordered:
            assert res["iter"] == it
            assert same_bits(res["rhistory"], GT[key + "/rhistory"]) and same_bits(res["x"], GT[key + "/x"])
        else:
            assert abs(res["iter"] - it) <= 1
            np.testing.assert_allclose(res["x"], GT[key + "/x"], rtol=1e-9, atol=1e-11)
        assert A.contents.matrix_type == capi.LIS_MATRIX_BSR and A.contents.is_splited == 1
        lib.lis_matrix_destroy(A)


@pytest.mark.parametrize("action", [1, 2])
def test_scaling_a_matrix_that_lives_in_hbm_only(lib, action):
    """lis_matrix_scale on a matrix adopted from HBM arrays (no host copy exists): diagonal, d, the rows' values and b by kernels, the plan
    rebuilt -- the same b, d and products (bits) as the host routine leaves on a host-resident copy of the same matrix (src/matrix/lis_matrix_ops.c:579-700)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_scale import test_matrix
    from lis_amd import DeviceArray as DA
    ptr, idx, val = test_matrix(257, 9)
    n = len(ptr) - 1
    x = np.sin(np.arange(n) * 0.7) + 0.25
    b0 = np.cos(np.arange(n) * 0.3) + 2.0
    # host-resident twin
    H = lisdrv.make_csr(lib, ptr, idx, val)
    hb, hd = lisdrv.new_vector(lib, H, b0), lisdrv.new_vector(lib, H)
    assert lib.lis_matrix_scale(H, hb, hd, action) == 0
    want_y, want_b, want_d = lisdrv.matvec(lib, H, x), lisdrv.get_vector(lib, hb, n), lisdrv.get_vector(lib, hd, n)
    # HBM-only matrix: arrays uploaded by hand and adopted (freed with the matrix)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    A = capi.PM()
    assert lib.lis_matrix_create(0, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, n, 0) == 0
    fn = lib.dll.lis_amd_matrix_set_csr_device
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, capi.PM]
    assert fn(len(idx), n, dptr.ptr, didx.ptr, dval.ptr, A) == 0
    dptr.ptr = didx.ptr = dval.ptr = None                                  # owned by A now
    assert same_bits(lisdrv.matvec(lib, A, x), orc.spmv_csr(ptr, idx, val, x))
    vb, vd = lisdrv.new_vector(lib, A, b0), lisdrv.new_vector(lib, A)
    assert lib.lis_matrix_scale(A, vb, vd, action) == 0
    assert same_bits(lisdrv.get_vector(lib, vd, n), want_d) and same_bits(lisdrv.get_vector(lib, vb, n), want_b)
    assert same_bits(lisdrv.matvec(lib, A, x), want_y)
    assert A.contents.is_scaled == 1
    # a constant-coefficient matrix born in HBM loses its value records with the scaling (boundary rows scale differently) and keeps the right product
    P = capi.PM()
    N = 12
    assert lib.lis_matrix_create(0, C.byref(P)) == 0 and lib.lis_matrix_set_size(P, 0, N ** 3) == 0
    lib.dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
    assert lib.dll.lis_amd_matrix_poisson3d(P, N, N, N, 0) == 0
    pp, pi, pv = orc.poisson3d(N, N, N)
    Hp = lisdrv.make_csr(lib, pp, pi, pv)
    xb = np.sin(np.arange(N ** 3) * 0.7) + 0.25
    pb, pd, qb, qd = (lisdrv.new_vector(lib, P, np.ones(N ** 3)) for _ in range(4))
    assert lib.lis_matrix_scale(P, pb, pd, action) == 0 and lib.lis_matrix_scale(Hp, qb, qd, action) == 0
    assert same_bits(lisdrv.matvec(lib, P, xb), lisdrv.matvec(lib, Hp, xb))
    for m in (H, A, P, Hp):
        lib.lis_matrix_destroy(m)
