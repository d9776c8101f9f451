"""Matrix Market reader / writers of liblis_amd.so against the reference.

Host-only code: runs without a GPU.  The checker is (1) tests/golden/mm_golden.npz, produced by the
reference's own lis_input / lis_output on the files under tests/golden/mm/ (make_golden_mm.py), and
(2) when oracle/_ref is built (dev container only), the reference library run live on freshly
generated files.  Everything is compared bit for bit: CSR ptr/index/value (the in-row entry ORDER is
what fixes the bits of later SpMVs), b, x, and the bytes of the files the writers produce.
"""
import ctypes as C
import os

import numpy as np
import pytest

import lis_amd
import lisdrv
from lis_amd import _capi as capi

HERE = os.path.dirname(os.path.abspath(__file__))
MM = os.path.join(HERE, "golden", "mm")
G = np.load(os.path.join(HERE, "golden", "mm_golden.npz"))
FILES = sorted(k[:-4] for k in G.files if k.endswith("/err"))


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    return lib


def read_with(lib, path, want=None, vectors=True):
    A, b, x = capi.PM(), capi.PV(), capi.PV()
    assert lib.lis_matrix_create(0, C.byref(A)) == 0
    assert lib.lis_vector_create(0, C.byref(b)) == 0
    assert lib.lis_vector_create(0, C.byref(x)) == 0
    if want is not None:
        assert lib.lis_matrix_set_type(A, want) == 0
    err = lib.lis_input(A, b if vectors else None, x if vectors else None, path.encode())
    return err, A, b, x


def destroy(lib, A, b, x):
    lib.lis_matrix_destroy(A); lib.lis_vector_destroy(b); lib.lis_vector_destroy(x)


@pytest.mark.parametrize("name", FILES)
def test_reader_matches_reference_golden(lib, name, capfd):
    err, A, b, x = read_with(lib, os.path.join(MM, name))
    out = capfd.readouterr().out
    if int(G[name + "/err"][0]) != 0:                      # testmat3.mtx is complex: "Not real", LIS_ERR_FILE_IO
        assert err == int(G[name + "/err"][0]) == 6
        destroy(lib, A, b, x)
        return
    assert err == 0
    arrs = lisdrv.matrix_arrays(A)
    assert arrs["type"] == capi.LIS_MATRIX_CSR and A.contents.status == capi.LIS_MATRIX_CSR
    assert "matrix size = %d x %d" % (arrs["n"], arrs["n"]) in out          # lis_input_mm.c:731
    for k in ("ptr", "index", "value"):
        assert np.array_equal(arrs[k], G[f"{name}/{k}"]), k
    for tag, v in (("b", b), ("x", x)):
        if f"{name}/{tag}" in G.files:
            assert not lib.lis_vector_is_null(v)
            assert np.array_equal(lisdrv.get_vector(lib, v, arrs["n"]), G[f"{name}/{tag}"])
        else:
            assert lib.lis_vector_is_null(v)               # test1.c:123 depends on this
    destroy(lib, A, b, x)


@pytest.mark.parametrize("name", [f for f in FILES if int(G[f + "/err"][0]) == 0])
def test_writers_match_reference_bytes(lib, name, tmp_path, capfd):
    err, A, b, x = read_with(lib, os.path.join(MM, name))
    assert err == 0
    tmp = str(tmp_path / "out.mtx")
    assert lib.lis_output_matrix(A, 2, tmp.encode()) == 0
    assert np.array_equal(np.frombuffer(open(tmp, "rb").read(), np.uint8), G[f"{name}/out_matrix"])
    if f"{name}/b" in G.files:
        for fmt, tag in ((1, "plain"), (2, "mm"), (3, "lis")):
            assert lib.lis_output_vector(b, fmt, tmp.encode()) == 0
            assert np.array_equal(np.frombuffer(open(tmp, "rb").read(), np.uint8), G[f"{name}/out_b_{tag}"]), tag
        xnull = capi.PV()
        lib.lis_vector_create(0, C.byref(xnull))
        assert lib.lis_output(A, b, xnull, 2, tmp.encode()) == 0
        assert np.array_equal(np.frombuffer(open(tmp, "rb").read(), np.uint8), G[f"{name}/out_matrix_b"])
        lib.lis_vector_destroy(xnull)
    capfd.readouterr()
    destroy(lib, A, b, x)


def test_harwell_boeing_requested_as_csc(lib, capfd):
    """an .rua file holds CSC arrays: asking for CSC keeps them as read (lis_input_hb.c:447-462)"""
    err, A, b, x = read_with(lib, os.path.join(MM, "gen_hb.rua"), want=capi.LIS_MATRIX_CSC)
    assert err == 0 and A.contents.matrix_type == capi.LIS_MATRIX_CSC
    arrs = lisdrv.matrix_arrays(A)
    assert arrs["nnz"] == 202 and arrs["ptr"][1] == 4 and np.array_equal(arrs["ptr"][:4], [0, 4, 10, 13])
    capfd.readouterr()
    destroy(lib, A, b, x)


def test_roundtrips_text_and_binary(lib, tmp_path, capfd):
    """write -> read gives the same arrays back, for the text and the binary (MMB) form, with b and x."""
    err, A, b, x = read_with(lib, os.path.join(MM, "gen_symmetric_bx.mtx"))
    assert err == 0
    a0 = lisdrv.matrix_arrays(A)
    b0, x0 = lisdrv.get_vector(lib, b, a0["n"]), lisdrv.get_vector(lib, x, a0["n"])
    for fmt in (2, 8):
        path = str(tmp_path / f"rt{fmt}.mtx")
        assert lib.lis_output(A, b, x, fmt, path.encode()) == 0
        err, A2, b2, x2 = read_with(lib, path)
        assert err == 0
        a2 = lisdrv.matrix_arrays(A2)
        for k in ("ptr", "index", "value"):
            assert np.array_equal(a0[k], a2[k])
        assert np.array_equal(lisdrv.get_vector(lib, b2, a0["n"]), b0)
        assert np.array_equal(lisdrv.get_vector(lib, x2, a0["n"]), x0)
        destroy(lib, A2, b2, x2)
    capfd.readouterr()
    destroy(lib, A, b, x)


def test_vector_files(lib, tmp_path):
    v = capi.PV()
    lib.lis_vector_create(0, C.byref(v))
    assert lib.lis_input_vector(v, os.path.join(MM, "testvec0.mtx").encode()) == 0
    vals = lisdrv.get_vector(lib, v, v.contents.n)
    assert np.array_equal(vals, G["testvec0.mtx/v"])
    # each writer's file is read back by the sniffing reader (lis_input.c:212-226)
    for fmt in (1, 2, 3):
        path = str(tmp_path / f"v{fmt}.txt")
        assert lib.lis_output_vector(v, fmt, path.encode()) == 0
        w = capi.PV()
        lib.lis_vector_create(0, C.byref(w))
        assert lib.lis_input_vector(w, path.encode()) == 0
        assert w.contents.n == v.contents.n
        assert np.array_equal(lisdrv.get_vector(lib, w, w.contents.n), vals)
        lib.lis_vector_destroy(w)
    lib.lis_vector_destroy(v)


@pytest.mark.parametrize("fmt", ["ell", "dia"])
def test_requested_storage_type_is_honoured(lib, fmt, capfd):
    """lis_matrix_set_type before lis_input: the matrix arrives converted (lis_input_mm.c:82-107)."""
    want = {"ell": capi.LIS_MATRIX_ELL, "dia": capi.LIS_MATRIX_DIA}[fmt]
    err, A, b, x = read_with(lib, os.path.join(MM, "testmat.mtx"), want=want)
    assert err == 0
    arrs = lisdrv.matrix_arrays(A)
    assert arrs["type"] == want
    for k, v in arrs.items():
        ref = G[f"testmat.mtx/{fmt}/{k}"]
        assert np.array_equal(np.atleast_1d(v), ref), k
    capfd.readouterr()
    destroy(lib, A, b, x)


def test_error_paths(lib, tmp_path, capfd):
    A = capi.PM()
    lib.lis_matrix_create(0, C.byref(A))
    assert lib.lis_input_matrix(A, str(tmp_path / "missing.mtx").encode()) == 6          # LIS_ERR_FILE_IO
    p = tmp_path / "rect.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 4 1\n1 1 1.0\n")
    assert lib.lis_input_matrix(A, str(p).encode()) == 6                                  # "matrix is not square"
    p = tmp_path / "pattern.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 1\n1 1\n")
    assert lib.lis_input_matrix(A, str(p).encode()) == 6                                  # "Not real"
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n")
    B = capi.PM()
    lib.lis_matrix_create(0, C.byref(B))
    assert lib.lis_input_matrix(B, str(p).encode()) == 6                                  # truncated file
    p = tmp_path / "hb.rua"                              # anything that is not Matrix Market is parsed as Harwell-Boeing
    p.write_text("%-72s%-8s\n%14d%14d%14d%14d\n%-14s%14d%14d%14d%14d\n" % ("title", "KEY", 3, 1, 1, 1, "RSA", 2, 2, 2, 0))
    Cm = capi.PM()
    lib.lis_matrix_create(0, C.byref(Cm))
    assert lib.lis_input_matrix(Cm, str(p).encode()) == 6                                 # "Not unsymmetric" (RUA only)
    assert lib.lis_input_matrix(A, None) == 1
    cap = capfd.readouterr()
    err = cap.out + cap.err                              # diagnostics go where the reference's lis_error sends them
    assert "not square" in err and "Not real" in err and "Not unsymmetric" in err
    for m in (A, B, Cm):
        lib.lis_matrix_destroy(m)


def test_indices_outside_the_matrix_are_refused(lib, tmp_path, capfd):
    """an out-of-range row / column would become a gather index of the HIP kernels: the readers fail with LIS_ERR_FILE_IO"""
    cases = {
        "col.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n2 4 5.0\n",
        "row.mtx": "%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 1.0\n7 2 5.0\n",
        "zero.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n0 1 1.0\n2 2 5.0\n",
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_text(text)
        A = capi.PM()
        lib.lis_matrix_create(0, C.byref(A))
        assert lib.lis_input_matrix(A, str(p).encode()) == 6, name
        lib.lis_matrix_destroy(A)
    # Harwell-Boeing: a generated file with one row index pushed out of range, and one that ends before its last card
    good = open(os.path.join(MM, "gen_hb.rua")).read().splitlines()
    head = good[:4]
    ptrcrd, indcrd = int(good[1][14:28]), int(good[1][28:42])
    bad = list(good)
    first_ind = 4 + ptrcrd
    bad[first_ind] = "%8d" % 99999 + bad[first_ind][8:]
    p = tmp_path / "badrow.rua"
    p.write_text("\n".join(bad) + "\n")
    A = capi.PM()
    lib.lis_matrix_create(0, C.byref(A))
    assert lib.lis_input_matrix(A, str(p).encode()) == 6
    lib.lis_matrix_destroy(A)
    p = tmp_path / "short.rua"
    p.write_text("\n".join(good[:4 + ptrcrd + indcrd]) + "\n")           # no value cards at all
    A = capi.PM()
    lib.lis_matrix_create(0, C.byref(A))
    assert lib.lis_input_matrix(A, str(p).encode()) == 6
    lib.lis_matrix_destroy(A)
    capfd.readouterr()


def test_live_against_reference_on_random_files(lib, reflib, tmp_path, capfd):
    """Dev container only: random general / symmetric files, the reference and this library read the same bytes."""
    rng = np.random.default_rng(7)
    for case in range(6):
        n = int(rng.integers(1, 60))
        sym = bool(case % 2)
        m = int(rng.integers(0, 5 * n))
        lines = []
        for _ in range(m):
            r, c = int(rng.integers(n)), int(rng.integers(n))
            if sym and c > r:
                r, c = c, r
            lines.append(f"{r + 1} {c + 1} {float(rng.normal()):.17g}")
        isb, isx = int(rng.integers(2)), int(rng.integers(2))
        path = str(tmp_path / f"r{case}.mtx")
        with open(path, "w") as f:
            f.write("%%%%MatrixMarket matrix coordinate real %s\n%% c\n" % ("symmetric" if sym else "general"))
            f.write(f"{n} {n} {m} {isb} {isx}\n" + "".join(l + "\n" for l in lines))
            for flag in (isb, isx):
                if flag:
                    f.write("".join(f"{i + 1} {float(rng.normal()):.17g}\n" for i in range(n)))
        e1, A1, b1, x1 = read_with(reflib, path)
        e2, A2, b2, x2 = read_with(lib, path)
        assert e1 == e2 == 0
        a1, a2 = lisdrv.matrix_arrays(A1), lisdrv.matrix_arrays(A2)
        for k in ("ptr", "index", "value"):
            assert np.array_equal(a1[k], a2[k]), (case, k)
        for v1, v2 in ((b1, b2), (x1, x2)):
            assert bool(reflib.lis_vector_is_null(v1)) == bool(lib.lis_vector_is_null(v2))
            if not lib.lis_vector_is_null(v2):
                assert np.array_equal(lisdrv.get_vector(reflib, v1, n), lisdrv.get_vector(lib, v2, n))
        o1, o2 = str(tmp_path / "o1"), str(tmp_path / "o2")
        assert reflib.lis_output_matrix(A1, 2, o1.encode()) == 0 and lib.lis_output_matrix(A2, 2, o2.encode()) == 0
        assert open(o1, "rb").read() == open(o2, "rb").read()
        destroy(reflib, A1, b1, x1); destroy(lib, A2, b2, x2)
    capfd.readouterr()


# ---------------------------------------------------------------- BASELINE config 4: the Queen_4147 stand-in through the reader
def test_queen_class_mini_reader_has_the_reference_bits(lib):
    """The generator of the Queen-scale case at a size the CPU suite can afford (5 184 rows, 311 K non-zeros): a SYMMETRIC coordinate
    file whose rows list their entries unsorted, node numbering scrambled.  This library's reader (parallel parse, exact-decimal
    fast path, counting sort by destination row) must leave the CSR arrays the reference's reader left
    (lis_input_mm.c:699-1069, expansion :986-1036) -- tests/golden/queen_class_golden.json, made from oracle/_ref -- and the oracle
    must reproduce the reference's product bits and iteration counts on them."""
    import hashlib
    import json
    import orc
    import queen_class
    g = json.load(open(os.path.join(HERE, "golden", "queen_class_golden.json")))["mini"]
    path, rows, stored = queen_class.generate("mini")
    try:
        assert os.path.getsize(path) == g["file_bytes"], "gen_queen_class.c drifted from the fixture"
        err, A, b, x = read_with(lib, path)
    finally:
        os.unlink(path)
    assert err == 0
    a = A.contents
    assert (a.n, a.nnz, rows, stored) == (g["n"], g["nnz"], g["n"], g["stored_entries"])
    ptr, idx, val = (np.ctypeslib.as_array(p, shape=(c,)).copy() for p, c in ((a.ptr, a.n + 1), (a.index, a.nnz), (a.value, a.nnz)))
    h = hashlib.sha256()
    for arr in (ptr, idx, val):
        h.update(arr.tobytes())
    assert h.hexdigest() == g["csr_sha256"]
    assert np.diff(ptr).max() == g["max_row"]
    assert len(np.unique(idx - np.repeat(np.arange(a.n), np.diff(ptr)))) > 255          # no diagonal structure to code
    n = a.n
    y = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.01) + 1.25)
    assert hashlib.sha256(y.tobytes()).hexdigest() == g["y_sha256"]
    rhs = orc.spmv_csr(ptr, idx, val, np.ones(n))
    for opts, want in g["solves"].items():
        tok = opts.split()
        solver, precon = tok[1], tok[tok.index("-p") + 1]
        kw = {"restart": 30} if solver == "gmres" else {}
        _, it, rc, resid, rh = getattr(orc, solver)(ptr, idx, val, rhs, precon=precon, tol=1e-12, maxiter=2000, **kw)
        assert (it, rc, resid) == (want["iter"], want["status"], want["resid"]), opts
        assert list(rh[:6]) == want["rhistory_head"], opts
    lib.lis_matrix_destroy(A)


def test_reader_reports_the_first_bad_line_of_a_large_file(lib, tmp_path, capfd):
    """the entry lines are parsed by several threads: the error reported is still the first one in FILE order, whichever piece holds it"""
    n, m = 3000, 400000
    rng = np.random.default_rng(5)
    r, c = rng.integers(1, n + 1, m), rng.integers(1, n + 1, m)
    lines = [f"{a} {b} {float(v)!r}" for a, b, v in zip(r, c, rng.uniform(-1, 1, m))]
    good = "%%MatrixMarket matrix coordinate real general\n" + f"{n} {n} {m}\n" + "\n".join(lines) + "\n"
    assert len(good) > (4 << 20)                     # beyond the single-thread threshold
    p = tmp_path / "big.mtx"
    p.write_text(good)
    err, A, _, _ = read_with(lib, str(p), vectors=False)
    assert err == 0 and A.contents.nnz == m
    ptr = np.ctypeslib.as_array(A.contents.ptr, shape=(n + 1,))
    idx = np.ctypeslib.as_array(A.contents.index, shape=(m,))
    order = np.argsort(r, kind="stable")             # file order inside every row
    assert np.array_equal(np.diff(ptr), np.bincount(r - 1, minlength=n)) and np.array_equal(idx, c[order] - 1)
    val = np.ctypeslib.as_array(A.contents.value, shape=(m,))
    assert np.array_equal(val, np.array([float(t.split()[2]) for t in lines])[order])          # Python's float() is correctly rounded too
    lib.lis_matrix_destroy(A)
    for k_bad, text in ((m - 7, f"{n + 1} 1 0.5"), (m // 2, "12 abc 1.0"), (5, "3 4")):
        bad = list(lines)
        bad[k_bad] = text
        bad[m - 3] = "0 0 1.0"                       # a later error must not win
        p.write_text("%%MatrixMarket matrix coordinate real general\n" + f"{n} {n} {m}\n" + "\n".join(bad) + "\n")
        capfd.readouterr()
        err, A, _, _ = read_with(lib, str(p), vectors=False)
        assert err == 6                              # LIS_ERR_FILE_IO
        said = "".join(capfd.readouterr())
        assert "entry %d:" % (m - 2) not in said
        if text.startswith(str(n + 1)):
            assert "entry %d: index (%d,1) is outside the matrix" % (k_bad + 1, n + 1) in said
        lib.lis_matrix_destroy(A)
    p.write_text("%%MatrixMarket matrix coordinate real general\n" + f"{n} {n} {m + 1}\n" + "\n".join(lines) + "\n")      # one line short
    err, A, _, _ = read_with(lib, str(p), vectors=False)
    assert err == 6
    lib.lis_matrix_destroy(A)
