"""Pin the CPU oracle (oracle/lis_oracle.c) bit-for-bit against the reference itself.

The reference is oracle/_ref/liblis_ref.so: Lis 2.1.11 compiled from /root/reference/src by
oracle/Makefile, driven through its own C API at 1 OpenMP thread.  No GPU involved.
"""
import numpy as np
import pytest

import lisdrv
import orc


def _cases():
    yield "p1d_100", orc.poisson1d(100)
    yield "p3d_6x5x4", orc.poisson3d(6, 5, 4)
    yield "p3d_8_sorted", orc.poisson3d(8, 8, 8, sort_cols=True)
    yield "rand_257", orc.random_csr(257, 9, seed=1)
    yield "rand_64_long", orc.random_csr(64, 5, seed=2, long_row=60)
    yield "rand_sorted_301", orc.random_csr(301, 6, seed=3, sort_cols=True)


CASES = list(_cases())
IDS = [c[0] for c in CASES]


def _x(n, seed=7):
    return np.random.default_rng(seed).uniform(-1, 1, n)


@pytest.mark.parametrize("name,csr", CASES, ids=IDS)
def test_spmv_csr(reflib, name, csr):
    ptr, idx, val = csr
    x = _x(len(ptr) - 1)
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    y_ref = lisdrv.matvec(reflib, A, x)
    assert np.array_equal(orc.spmv_csr(ptr, idx, val, x), y_ref)
    reflib.lis_matrix_destroy(A)


@pytest.mark.parametrize("name,csr", CASES, ids=IDS)
@pytest.mark.parametrize("fmt", ["csc", "ell", "dia", "jad", "bsr"])
def test_convert_and_spmv(reflib, name, csr, fmt):
    ptr, idx, val = csr
    n = len(ptr) - 1
    x = _x(n)
    # csr2dia sorts its INPUT in place (lis_matrix_dia.c:1217): convert from a private copy
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    B = lisdrv.convert(reflib, A, fmt)
    arrs = lisdrv.matrix_arrays(B)
    y_ref = lisdrv.matvec(reflib, B, x)
    if fmt == "csc":
        cptr, cidx, cval = orc.csr2csc(ptr, idx, val)
        assert np.array_equal(cptr, arrs["ptr"])
        assert np.array_equal(cidx, arrs["index"])
        assert np.array_equal(cval, arrs["value"])
        y = orc.spmv_csc(n, n, cptr, cidx, cval, x)
    elif fmt == "ell":
        mx, eidx, eval_ = orc.csr2ell(ptr, idx, val)
        assert mx == arrs["maxnzr"]
        assert np.array_equal(eidx, arrs["index"])
        assert np.array_equal(eval_, arrs["value"])
        y = orc.spmv_ell(n, mx, eidx, eval_, x)
    elif fmt == "dia":
        sidx, sval = orc.sort_rows(ptr, idx, val)
        nnd, off, dval = orc.csr2dia(ptr, sidx, sval)
        assert nnd == arrs["nnd"]
        assert np.array_equal(off, arrs["index"])
        assert np.array_equal(dval, arrs["value"])
        y = orc.spmv_dia(n, nnd, off, dval, x)
    elif fmt == "jad":
        mx, perm, jptr, jidx, jval = orc.csr2jad(ptr, idx, val)
        assert mx == arrs["maxnzr"]
        assert np.array_equal(perm, arrs["row"])
        assert np.array_equal(jptr, arrs["ptr"])
        assert np.array_equal(jidx, arrs["index"])
        assert np.array_equal(jval, arrs["value"])
        y = orc.spmv_jad(n, mx, perm, jptr, jidx, jval, x)
    else:
        nr, bptr, bidx, bval = orc.csr2bsr(ptr, idx, val, 2, 2)
        assert nr == arrs["nr"]
        assert np.array_equal(bptr, arrs["bptr"])
        assert np.array_equal(bidx, arrs["bindex"])
        assert np.array_equal(bval, arrs["value"])
        y = orc.spmv_bsr(n, nr, 2, 2, bptr, bidx, bval, x)
    assert np.array_equal(y, y_ref), fmt
    reflib.lis_matrix_destroy(A)
    reflib.lis_matrix_destroy(B)


@pytest.mark.parametrize("name,csr", CASES, ids=IDS)
@pytest.mark.parametrize("fmt", ["csr", "csc", "ell", "dia", "jad", "bsr"])
def test_transposed_spmv(reflib, name, csr, fmt):
    """lis_matvech of the reference (1 thread) == the oracle's scatter restatement, every format."""
    ptr, idx, val = csr
    n = len(ptr) - 1
    x = _x(n, seed=9)
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(reflib, A, fmt)
    y_ref = lisdrv.matvech(reflib, B, x)
    if fmt == "csr":
        y = orc.spmvh_csr(ptr, idx, val, x)
    elif fmt == "csc":
        cptr, cidx, cval = orc.csr2csc(ptr, idx, val)
        y = orc.spmvh_csc(n, cptr, cidx, cval, x)
    elif fmt == "ell":
        mx, eidx, eval_ = orc.csr2ell(ptr, idx, val)
        y = orc.spmvh_ell(n, mx, eidx, eval_, x)
    elif fmt == "dia":
        sidx, sval = orc.sort_rows(ptr, idx, val)
        nnd, off, dval = orc.csr2dia(ptr, sidx, sval)
        y = orc.spmvh_dia(n, nnd, off, dval, x)
    elif fmt == "jad":
        mx, perm, jptr, jidx, jval = orc.csr2jad(ptr, idx, val)
        y = orc.spmvh_jad(n, mx, perm, jptr, jidx, jval, x)
    else:
        nr, bptr, bidx, bval = orc.csr2bsr(ptr, idx, val, 2, 2)
        y = orc.spmvh_bsr(n, nr, 2, 2, bptr, bidx, bval, x)
    assert np.array_equal(y, y_ref), fmt
    if B is not A:
        reflib.lis_matrix_destroy(B)
    reflib.lis_matrix_destroy(A)


@pytest.mark.parametrize("bnr,bnc", [(1, 1), (2, 3), (3, 2), (4, 4), (3, 3)])
def test_bsr_blocks(reflib, bnr, bnc):
    ptr, idx, val = orc.random_csr(120, 7, seed=11)
    n = 120
    x = _x(n)
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    B = lisdrv.convert(reflib, A, "bsr", bnr, bnc)
    arrs = lisdrv.matrix_arrays(B)
    nr, bptr, bidx, bval = orc.csr2bsr(ptr, idx, val, bnr, bnc)
    assert np.array_equal(bptr, arrs["bptr"]) and np.array_equal(bidx, arrs["bindex"])
    assert np.array_equal(bval, arrs["value"])
    assert np.array_equal(orc.spmv_bsr(n, nr, bnr, bnc, bptr, bidx, bval, x),
                          lisdrv.matvec(reflib, B, x))


def test_vector_ops(reflib):
    import ctypes as C
    n = 1003
    rng = np.random.default_rng(5)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    ptr, idx, val = orc.poisson1d(n)
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    vx, vy, vz = (lisdrv.new_vector(reflib, A, v) for v in (x, y, np.zeros(n)))
    out = C.c_double()
    reflib.lis_vector_dot(vx, vy, C.byref(out))
    assert out.value == orc.lib().orc_dot(n, x, y)
    reflib.lis_vector_nrm2(vx, C.byref(out))
    assert out.value == orc.lib().orc_nrm2(n, x)
    reflib.lis_vector_nrm1(vx, C.byref(out))
    assert out.value == orc.lib().orc_nrm1(n, x)
    a = 0.3712
    yy = y.copy(); orc.lib().orc_axpy(n, a, x, yy)
    reflib.lis_vector_axpy(a, vx, vy)
    assert np.array_equal(lisdrv.get_vector(reflib, vy), yy)
    orc.lib().orc_xpay(n, x, a, yy)
    reflib.lis_vector_xpay(vx, a, vy)
    assert np.array_equal(lisdrv.get_vector(reflib, vy), yy)
    zz = np.empty(n); orc.lib().orc_axpyz(n, a, x, yy, zz)
    reflib.lis_vector_axpyz(a, vx, vy, vz)
    assert np.array_equal(lisdrv.get_vector(reflib, vz), zz)
    orc.lib().orc_scale(n, a, zz)
    reflib.lis_vector_scale(a, vz)
    assert np.array_equal(lisdrv.get_vector(reflib, vz), zz)
    orc.lib().orc_pmul(n, x, yy, zz)
    reflib.lis_vector_pmul(vx, vy, vz)
    assert np.array_equal(lisdrv.get_vector(reflib, vz), zz)
    orc.lib().orc_reciprocal(n, zz)
    reflib.lis_vector_reciprocal(vz)
    assert np.array_equal(lisdrv.get_vector(reflib, vz), zz)
    d = orc.csr_diagonal(ptr, idx, val)
    reflib.lis_matrix_get_diagonal(A, vz)
    assert np.array_equal(lisdrv.get_vector(reflib, vz), d)


SOLVER_CASES = [
    ("cg", "jacobi", (8, 8, 8)), ("cg", "none", (6, 7, 5)),
    ("bicgstab", "none", (8, 8, 8)), ("bicgstab", "jacobi", (5, 6, 7)),
    ("gmres", "none", (8, 8, 8)), ("gmres", "jacobi", (6, 6, 6)),
    ("bicg", "none", (8, 8, 8)), ("bicg", "jacobi", (5, 6, 7)),
]


@pytest.mark.parametrize("solver,precon,grid", SOLVER_CASES)
def test_solvers_poisson(reflib, solver, precon, grid):
    ptr, idx, val = orc.poisson3d(*grid)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.ones(n))
    A = lisdrv.make_csr(reflib, ptr, idx, val)
    opts = f"-i {solver} -p {precon} -tol 1e-12 -maxiter 500 -print mem"
    extra = {}
    if solver == "gmres":
        opts += " -restart 7"
        extra["restart"] = 7
    ref = lisdrv.solve(reflib, A, b, opts)
    x, it, rc, resid, rh = getattr(orc, solver)(ptr, idx, val, b, precon=precon, tol=1e-12,
                                                maxiter=500, **extra)
    assert it == ref["iter"] and rc == ref["status"]
    assert resid == ref["resid"]
    assert np.array_equal(x, ref["x"])
    assert np.array_equal(rh[1:it + 1], ref["rhistory"][1:it + 1])


def test_solver_nonsymmetric_and_maxiter(reflib):
    ptr, idx, val = orc.random_csr(200, 6, seed=21, empty_rows=False)
    # make it diagonally dominant so the iterations behave
    n = 200
    dense_diag = np.zeros(n)
    for r in range(n):
        dense_diag[r] = np.abs(val[ptr[r]:ptr[r + 1]]).sum() + 1.0
    # append the diagonal as a last entry per row
    nptr = ptr + np.arange(n + 1, dtype=np.int32)
    nidx = np.empty(len(idx) + n, np.int32)
    nval = np.empty(len(idx) + n)
    for r in range(n):
        s, e = ptr[r], ptr[r + 1]
        cols, vals = idx[s:e], val[s:e].copy()
        vals[cols == r] = 0.0
        nidx[nptr[r]:nptr[r + 1] - 1] = cols
        nval[nptr[r]:nptr[r + 1] - 1] = vals
        nidx[nptr[r + 1] - 1] = r
        nval[nptr[r + 1] - 1] = dense_diag[r]
    b = orc.spmv_csr(nptr, nidx, nval, np.ones(n))
    A = lisdrv.make_csr(reflib, nptr, nidx, nval)
    for solver, extra, o in (("bicgstab", {}, ""), ("gmres", {"restart": 5}, " -restart 5"), ("bicg", {}, "")):
        for maxiter in (3, 400):
            ref = lisdrv.solve(reflib, A, b, f"-i {solver} -p none -maxiter {maxiter} -print mem" + o)
            x, it, rc, resid, rh = getattr(orc, solver)(nptr, nidx, nval, b, maxiter=maxiter, **extra)
            assert (it, rc) == (ref["iter"], ref["status"]), (solver, maxiter)
            assert resid == ref["resid"]
            assert np.array_equal(x, ref["x"])
