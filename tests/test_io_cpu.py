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
