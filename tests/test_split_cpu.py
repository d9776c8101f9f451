"""The split form A = L + D + U without a GPU: the oracle's restatement of the reference's split products against what the
reference returned (tests/golden/split_golden.npz, made by make_golden_split.py), and the arrays lis_matrix_split of
liblis_amd.so builds against the reference's (host code)."""
import os

import numpy as np
import pytest

import lis_amd
import lisdrv
import orc
from lis_amd import _capi as capi

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_golden.npz"))
MATS = ["p3d_6x5x4", "nonsym_61", "zeros_40"]
CASES = [("csr", 0), ("csc", 0), ("ell", 0), ("dia", 0), ("jad", 0), ("bsr", 1), ("bsr", 2), ("bsr", 3), ("bsr", 4), ("bsr", 5)]


def same_bits(a, b):
    return np.array_equal(np.asarray(a, np.float64).view(np.int64), np.asarray(b, np.float64).view(np.int64))


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    return lib


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", CASES)
def test_oracle_split_product_has_the_reference_bits(name, fmt, bs):
    ptr, idx, val, x = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val", "x"))
    n = len(ptr) - 1
    if fmt == "csr":
        y = orc.spmv_split_csr(ptr, idx, val, x)
    elif fmt == "csc":
        y = orc.spmv_split_csc(n, *orc.csr2csc(ptr, idx, val), x)
    elif fmt == "ell":
        y = orc.spmv_split_ell(n, *orc.csr2ell(ptr, idx, val), x)
    elif fmt == "dia":
        y = orc.spmv_split_dia(n, *orc.csr2dia(ptr, *orc.sort_rows(ptr, idx, val)), x)
    elif fmt == "jad":
        y = orc.spmv_split_jad(n, *orc.csr2jad(ptr, idx, val), x)
    else:
        nr, bptr, bidx, bval = orc.csr2bsr(ptr, idx, val, bs, bs)
        y = orc.spmv_split_bsr(n, nr, bs, bs, bptr, bidx, bval, x)
    assert same_bits(y, G[f"{name}/{fmt}{bs if bs else ''}/y_split"])          # signed zeros included


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", CASES)
def test_lis_matrix_split_builds_the_reference_arrays(lib, name, fmt, bs):
    ptr, idx, val = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val"))
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt, bs, bs)
    assert lib.lis_matrix_split(B) == 0 and B.contents.is_splited == 1
    assert lib.lis_matrix_split(B) == 0                                         # idempotent (lis_matrix_ops.c:866-870)
    parts = lisdrv.split_arrays(B)
    key = f"{name}/{fmt}{bs if bs else ''}"
    for tag in ("L", "U"):
        for k, v in parts[tag].items():
            want = G[f"{key}/{tag}/{k}"]
            assert (same_bits(v, want) if np.asarray(v).dtype == np.float64 else np.array_equal(v, want)), (tag, k)
    assert same_bits(parts["D"], G[key + "/D"])
    d = lisdrv.new_vector(lib, B)
    assert lib.lis_matrix_get_diagonal(B, d) == 0                               # reads D of the split form
    assert lib.lis_matrix_merge(B) == 0 and B.contents.is_splited == 0
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


def test_split_refuses_what_the_reference_refuses(lib, capfd):
    ptr, idx, val = (G[f"p3d_6x5x4/{k}"] for k in ("ptr", "idx", "val"))
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = lisdrv.convert(lib, A, "bsr", 3, 2)
    assert lib.lis_matrix_split(B) == capi.LIS_ERR_NOT_IMPLEMENTED              # non-square blocks (lis_matrix_bsr.c:1164)
    capfd.readouterr()
    lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


# ---------------------------------------------------------------- -scale jacobi -storage bsr (block-diagonal scaling)
GB = np.load(os.path.join(os.path.dirname(__file__), "golden", "bscale_golden.npz"))
BCASES = sorted({k.rsplit("/", 1)[0] for k in GB.files})


def bscale_matrix(name):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_scale import test_matrix
    if name == "p3d_7x6x5":
        return orc.poisson3d(7, 6, 5)
    if name == "p3d_odd_5x5x3":
        return orc.poisson3d(5, 5, 3)
    return test_matrix(61, 5)


@pytest.mark.parametrize("key", BCASES)
def test_block_scaling_leaves_the_reference_arrays(lib, key):
    name, _, blk = key.split("/")
    block = int(blk[1:])
    ptr, idx, val = bscale_matrix(name)
    n = len(ptr) - 1
    b = orc.spmv_csr(ptr, idx, val, np.cos(np.arange(n) * 0.3) + 2.0)
    A = lisdrv.make_csr(lib, ptr, idx, val)
    B = lisdrv.convert(lib, A, "bsr", block, block)
    vb = lisdrv.new_vector(lib, B, b)
    fn = lib.dll.lisi_matrix_bscale_bsr
    fn.argtypes = [capi.PM, capi.PV]
    assert fn(B, vb) == 0
    parts = lisdrv.split_arrays(B)
    assert same_bits(parts["L"]["value"], GB[key + "/L"]) and same_bits(parts["U"]["value"], GB[key + "/U"])
    assert same_bits(parts["D"], GB[key + "/D"])
    assert same_bits(np.ctypeslib.as_array(vb.contents.value, shape=(n,)), GB[key + "/b_scaled"])
    lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)


@pytest.mark.parametrize("name", MATS)
@pytest.mark.parametrize("fmt,bs", [("csr", 0), ("bsr", 2), ("bsr", 3)])
def test_lis_matrix_merge_rebuilds_the_arrays_from_the_parts(lib, name, fmt, bs):
    """lis_matrix_merge of a CSR / BSR matrix rebuilds A's arrays from L, D, U in that order with the parts' CURRENT values
    (lis_matrix_merge_csr, lis_matrix_merge_bsr: lis_matrix_bsr.c:1337-1395) -- a diagonal entry appears where the row had none --
    and lis_matrix_convert / lis_matrix_copy merge a split input first (lis_matrix_ops.c:142).  Checked live against the reference
    when oracle/_ref is built, against the construction rule otherwise."""
    ptr, idx, val = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val"))
    libs = [lib]
    if os.path.exists(orc.REF_SO):
        libs.append(lisdrv.open_lib(orc.REF_SO, threads=1))
    got = []
    for L in libs:
        A = lisdrv.make_csr(L, ptr, idx, val)
        B = A if fmt == "csr" else lisdrv.convert(L, A, fmt, bs, bs)
        assert L.lis_matrix_split(B) == 0
        parts = lisdrv.split_arrays(B)
        # scale the parts in place, as -scale jacobi -storage bsr does: the merged arrays must carry the scaled values
        for core in (B.contents.L.contents, B.contents.U.contents):
            cnt = core.nnz if fmt == "csr" else core.bnnz * bs * bs
            if cnt:
                np.ctypeslib.as_array(core.value, shape=(cnt,))[:] *= 0.5
        assert L.lis_matrix_merge(B) == 0 and B.contents.is_splited == 0
        arrs = lisdrv.matrix_arrays(B)
        got.append(arrs)
        pk, ik = ("ptr", "index") if fmt == "csr" else ("bptr", "bindex")
        blk = 1 if fmt == "csr" else bs * bs
        rows = len(arrs[pk]) - 1
        lp, up = parts["L"][pk], parts["U"][pk]
        want_ptr = lp + up + np.arange(rows + 1)
        assert np.array_equal(arrs[pk], want_ptr)
        for r in range(rows):
            a = want_ptr[r]
            nl, nu = lp[r + 1] - lp[r], up[r + 1] - up[r]
            assert np.array_equal(arrs[ik][a:a + nl], parts["L"][ik][lp[r]:lp[r + 1]]) and arrs[ik][a + nl] == r
            assert np.array_equal(arrs[ik][a + nl + 1:a + nl + 1 + nu], parts["U"][ik][up[r]:up[r + 1]])
            assert same_bits(arrs["value"][a * blk:(a + nl) * blk], 0.5 * parts["L"]["value"][lp[r] * blk:lp[r + 1] * blk])
            assert same_bits(arrs["value"][(a + nl) * blk:(a + nl + 1) * blk], parts["D"][r * blk:(r + 1) * blk])
            assert same_bits(arrs["value"][(a + nl + 1) * blk:(a + nl + 1 + nu) * blk], 0.5 * parts["U"]["value"][up[r] * blk:up[r + 1] * blk])
        # a split input is merged by convert before anything is read from it
        assert L.lis_matrix_split(B) == 0
        Cc = lisdrv.convert(L, B, "csr")
        assert B.contents.is_splited == 0
        got.append(lisdrv.matrix_arrays(Cc))
        L.lis_matrix_destroy(Cc)
        if B is not A:
            L.lis_matrix_destroy(B)
        L.lis_matrix_destroy(A)
    if len(libs) == 2:
        for mine, ref in ((got[0], got[2]), (got[1], got[3])):
            for k in mine:
                if isinstance(mine[k], np.ndarray):
                    assert np.array_equal(mine[k], ref[k]) and (mine[k].dtype != np.float64 or same_bits(mine[k], ref[k])), k


# ---------------------------------------------------------------- round 4: scaling a split matrix is host code (no GPU needed for the arrays)
GT = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_t_golden.npz"))


@pytest.mark.parametrize("name", ["p3d_6x5x4", "nonsym_61", "zeros_40"])
@pytest.mark.parametrize("fmt,bs", [("csr", 0), ("bsr", 2), ("bsr", 3)])
@pytest.mark.parametrize("action", [1, 2])
def test_scaling_a_split_matrix_on_the_host(lib, name, fmt, bs, action):
    """lis_matrix_scale on A = L + D + U: the parts and b, d as the reference leaves them (lis_matrix_csr.c:617-632, :661-676: D becomes 1;
    lis_matrix_bsr.c:820-855, :895-935: the diagonal blocks scaled by d[row]*d[row] in the symmetric case, as there)"""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_golden.npz"))
    ptr, idx, val = (G[f"{name}/{k}"] for k in ("ptr", "idx", "val"))
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
    lib.lis_vector_destroy(vb); lib.lis_vector_destroy(vd)
    if B is not A:
        lib.lis_matrix_destroy(B)
    lib.lis_matrix_destroy(A)
