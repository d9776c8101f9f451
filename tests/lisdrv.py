"""Drive a library exporting the Lis C API (the reference build or this repo's liblis_amd.so) from numpy.

The calls below are the ones the reference's own drivers make (test/spmvtest*.c, test/test3.c):
create -> set_size -> set_csr -> assemble -> duplicate/set_type/convert -> lis_matvec / lis_solve.
"""
import ctypes as C

import numpy as np

from lis_amd import _capi as capi

_initialized = {}


def open_lib(path, threads=1):
    """Load + lis_initialize once per path.  threads only matters for the OpenMP reference build."""
    if path in _initialized:
        return _initialized[path]
    lib = capi.LisLib(path)
    err = lib.initialize(["-omp_num_threads", str(threads)])
    assert err == 0
    _initialized[path] = lib
    return lib


def _copy_in(dst_ptr, arr):
    C.memmove(dst_ptr, arr.ctypes.data, arr.nbytes)


def _arr(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)


def make_csr(lib, ptr, idx, val, n=None, gn=0):
    """Rows given by ptr.  gn=0: set_size(A, rows, 0) (local size given); gn>0 with n=0: set_size(A, 0, gn), the
    drivers' call pattern, where the library splits the global rows over the ranks (LIS_GET_ISIE)."""
    rows = len(ptr) - 1
    A = capi.PM()
    assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0
    assert lib.lis_matrix_set_size(A, rows if n is None else n, gn) == 0
    n = rows
    p, i, v = capi.P_INT(), capi.P_INT(), capi.P_DBL()
    nnz = int(ptr[-1])
    assert lib.lis_matrix_malloc_csr(n, max(nnz, 1), C.byref(p), C.byref(i), C.byref(v)) == 0
    _copy_in(p, np.ascontiguousarray(ptr, np.int32))
    if nnz:
        _copy_in(i, np.ascontiguousarray(idx, np.int32))
        _copy_in(v, np.ascontiguousarray(val, np.float64))
    assert lib.lis_matrix_set_csr(nnz, p, i, v, A) == 0
    assert lib.lis_matrix_assemble(A) == 0
    return A


def convert(lib, A, fmt, bnr=2, bnc=2):
    B = capi.PM()
    assert lib.lis_matrix_duplicate(A, C.byref(B)) == 0
    assert lib.lis_matrix_set_type(B, capi.FORMAT_ID[fmt]) == 0
    if fmt == "bsr":
        assert lib.lis_matrix_set_blocksize(B, bnr, bnc, None, None) == 0
    err = lib.lis_matrix_convert(A, B)
    assert err == 0, err
    return B


def new_vector(lib, A, values=None):
    v = capi.PV()
    assert lib.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(v)) == 0
    if values is not None:
        set_vector(lib, v, values)
    return v


def set_vector(lib, v, values):
    values = np.ascontiguousarray(values, np.float64)
    assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, 0, len(values),
                                      values.ctypes.data_as(capi.P_DBL), v) == 0


def get_vector(lib, v, count=None):
    n = v.contents.n if count is None else count
    out = np.empty(n)
    assert lib.lis_vector_get_values(v, 0, n, out.ctypes.data_as(capi.P_DBL)) == 0
    return out


def matvec(lib, A, x):
    vx = new_vector(lib, A, x)
    vy = new_vector(lib, A)
    err = lib.lis_matvec(A, vx, vy)
    assert err == 0, err
    y = get_vector(lib, vy, A.contents.n)
    lib.lis_vector_destroy(vx)
    lib.lis_vector_destroy(vy)
    return y


def matvech(lib, A, x):
    """y = A^T x through lis_matvech (ref src/matvec/lis_matvec.c:191)."""
    vx = new_vector(lib, A, x)
    vy = new_vector(lib, A)
    err = lib.lis_matvech(A, vx, vy)
    assert err == 0, err
    y = get_vector(lib, vy, A.contents.n)
    lib.lis_vector_destroy(vx)
    lib.lis_vector_destroy(vy)
    return y


def matrix_arrays(A):
    """Host arrays of an assembled matrix, by format (layouts: SURVEY 8a rows a5-a11)."""
    a = A.contents
    t = a.matrix_type
    n, np_ = a.n, a.np
    out = {"type": t, "n": n, "np": np_, "nnz": a.nnz}
    if t == capi.LIS_MATRIX_CSR:
        out.update(ptr=_arr(a.ptr, n + 1, np.int32), index=_arr(a.index, a.nnz, np.int32),
                   value=_arr(a.value, a.nnz, np.float64))
    elif t == capi.LIS_MATRIX_CSC:
        out.update(ptr=_arr(a.ptr, np_ + 1, np.int32), index=_arr(a.index, a.nnz, np.int32),
                   value=_arr(a.value, a.nnz, np.float64))
    elif t == capi.LIS_MATRIX_ELL:
        out.update(maxnzr=a.maxnzr, index=_arr(a.index, a.maxnzr * n, np.int32),
                   value=_arr(a.value, a.maxnzr * n, np.float64))
    elif t == capi.LIS_MATRIX_DIA:
        out.update(nnd=a.nnd, index=_arr(a.index, a.nnd, np.int32),
                   value=_arr(a.value, a.nnd * n, np.float64))
    elif t == capi.LIS_MATRIX_JAD:
        out.update(maxnzr=a.maxnzr, row=_arr(a.row, n, np.int32),
                   ptr=_arr(a.ptr, a.maxnzr + 1, np.int32),      # first chunk (1 thread)
                   index=_arr(a.index, a.nnz, np.int32), value=_arr(a.value, a.nnz, np.float64))
    elif t == capi.LIS_MATRIX_BSR:
        bs = a.bnr * a.bnc
        out.update(bnr=a.bnr, bnc=a.bnc, nr=a.nr, nc=a.nc, bnnz=a.bnnz, pad=a.pad,
                   bptr=_arr(a.bptr, a.nr + 1, np.int32), bindex=_arr(a.bindex, a.bnnz, np.int32),
                   value=_arr(a.value, a.bnnz * bs, np.float64))
    return out


def solve(lib, A, b, options, x0=None):
    """lis_solve with `options` text; returns dict(x, iter, retcode, resid, rhistory, status)."""
    vb = new_vector(lib, A, b)
    vx = new_vector(lib, A, x0)
    S = capi.PS()
    assert lib.lis_solver_create(C.byref(S)) == 0
    assert lib.lis_solver_set_option(options.encode(), S) == 0
    err = lib.lis_solve(A, vb, vx, S)
    it = C.c_int()
    res = C.c_double()
    st = C.c_int()
    lib.lis_solver_get_iter(S, C.byref(it))
    lib.lis_solver_get_residualnorm(S, C.byref(res))
    lib.lis_solver_get_status(S, C.byref(st))
    maxiter = S.contents.options[2]
    rh = _arr(S.contents.rhistory, min(it.value, maxiter) + 1, np.float64) \
        if S.contents.rhistory else np.zeros(0)
    out = dict(err=err, x=get_vector(lib, vx, A.contents.n), iter=it.value, resid=res.value,
               status=st.value, rhistory=rh, itime=S.contents.itime, time=S.contents.time)
    lib.lis_solver_destroy(S)
    lib.lis_vector_destroy(vb)
    lib.lis_vector_destroy(vx)
    return out


def split_arrays(A):
    """L / U / D of a split matrix as numpy arrays, by storage format (layouts: src/matrix/lis_matrix_<fmt>.c split routines)."""
    a = A.contents
    assert a.is_splited
    t, n, np_ = a.matrix_type, a.n, a.np
    out = {}
    for tag, core in (("L", a.L.contents), ("U", a.U.contents)):
        if t in (capi.LIS_MATRIX_CSR, capi.LIS_MATRIX_CSC):
            lines = n if t == capi.LIS_MATRIX_CSR else np_
            out[tag] = dict(nnz=core.nnz, ptr=_arr(core.ptr, lines + 1, np.int32), index=_arr(core.index, core.nnz, np.int32),
                            value=_arr(core.value, core.nnz, np.float64))
        elif t == capi.LIS_MATRIX_ELL:
            out[tag] = dict(maxnzr=core.maxnzr, index=_arr(core.index, core.maxnzr * n, np.int32), value=_arr(core.value, core.maxnzr * n, np.float64))
        elif t == capi.LIS_MATRIX_DIA:
            out[tag] = dict(nnd=core.nnd, index=_arr(core.index, core.nnd, np.int32), value=_arr(core.value, core.nnd * n, np.float64))
        elif t == capi.LIS_MATRIX_JAD:
            out[tag] = dict(nnz=core.nnz, maxnzr=core.maxnzr, row=_arr(core.row, n, np.int32), ptr=_arr(core.ptr, core.maxnzr + 1, np.int32),
                            index=_arr(core.index, core.nnz, np.int32), value=_arr(core.value, core.nnz, np.float64))
        elif t == capi.LIS_MATRIX_BSR:
            bs = core.bnr * core.bnc
            out[tag] = dict(bnnz=core.bnnz, bptr=_arr(core.bptr, core.nr + 1, np.int32), bindex=_arr(core.bindex, core.bnnz, np.int32),
                            value=_arr(core.value, core.bnnz * bs, np.float64))
    dcount = a.nr * a.bnr * a.bnc if t == capi.LIS_MATRIX_BSR else n
    out["D"] = _arr(a.D.contents.value, dcount, np.float64)
    return out
