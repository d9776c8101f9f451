"""The CPU oracle against the committed golden vectors (tests/golden/lis_ref_golden.npz, produced from the
reference itself by tests/golden/make_golden.py).  Needs neither /root/reference nor a GPU."""
import os

import numpy as np
import pytest

import orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lis_ref_golden.npz"))
MATS = ["p1d100", "p3d_6x5x4", "p3d_8s", "irr150"]
SOLVES = sorted({k.split("/")[1] for k in G.files if k.startswith("solve/")})


def mat(name):
    return G[f"{name}/ptr"], G[f"{name}/idx"], G[f"{name}/val"], G[f"{name}/x"]


@pytest.mark.parametrize("name", MATS)
def test_spmv_all_formats(name):
    ptr, idx, val, x = mat(name)
    n = len(ptr) - 1
    assert np.array_equal(orc.spmv_csr(ptr, idx, val, x), G[f"{name}/y_csr"])
    cptr, cidx, cval = orc.csr2csc(ptr, idx, val)
    assert np.array_equal(cptr, G[f"{name}/csc/ptr"]) and np.array_equal(cidx, G[f"{name}/csc/index"])
    assert np.array_equal(orc.spmv_csc(n, n, cptr, cidx, cval, x), G[f"{name}/y_csc"])
    mx, eidx, ev = orc.csr2ell(ptr, idx, val)
    assert np.array_equal(eidx, G[f"{name}/ell/index"]) and np.array_equal(ev, G[f"{name}/ell/value"])
    assert np.array_equal(orc.spmv_ell(n, mx, eidx, ev, x), G[f"{name}/y_ell"])
    sidx, sval = orc.sort_rows(ptr, idx, val)
    nnd, off, dv = orc.csr2dia(ptr, sidx, sval)
    assert np.array_equal(off, G[f"{name}/dia/index"]) and np.array_equal(dv, G[f"{name}/dia/value"])
    assert np.array_equal(orc.spmv_dia(n, nnd, off, dv, x), G[f"{name}/y_dia"])
    mx, perm, jptr, jidx, jv = orc.csr2jad(ptr, idx, val)
    assert np.array_equal(perm, G[f"{name}/jad/row"]) and np.array_equal(jptr, G[f"{name}/jad/ptr"])
    assert np.array_equal(jidx, G[f"{name}/jad/index"]) and np.array_equal(jv, G[f"{name}/jad/value"])
    assert np.array_equal(orc.spmv_jad(n, mx, perm, jptr, jidx, jv, x), G[f"{name}/y_jad"])
    nr, bptr, bidx, bv = orc.csr2bsr(ptr, idx, val)
    assert np.array_equal(bptr, G[f"{name}/bsr/bptr"]) and np.array_equal(bidx, G[f"{name}/bsr/bindex"])
    assert np.array_equal(bv, G[f"{name}/bsr/value"])
    assert np.array_equal(orc.spmv_bsr(n, nr, 2, 2, bptr, bidx, bv, x), G[f"{name}/y_bsr"])


def test_known_answers_of_the_reference_drivers():
    # spmvtest1: ||A*1||_2 = sqrt(2) for the 1-D matrix (doc + SURVEY 8c)
    ptr, idx, val = orc.poisson1d(10000)
    y = orc.spmv_csr(ptr, idx, val, np.ones(10000))
    assert abs(orc.lib().orc_nrm2(10000, y) - np.sqrt(2.0)) < 1e-15
    # spmvtest3 N=32: ||A*1||_2^2 = 6(N-2)^2 + 12(N-2)*4 + 8*9
    ptr, idx, val = orc.poisson3d(32, 32, 32)
    y = orc.spmv_csr(ptr, idx, val, np.ones(32 ** 3))
    assert orc.lib().orc_nrm2(32 ** 3, y) == G["known/p3d_32/nrm2_A1"][0] == np.sqrt(6 * 900 + 12 * 30 * 4 + 72)


@pytest.mark.parametrize("name", SOLVES)
def test_solvers(name):
    solver, precon = name.split("_")[0], name.split("_")[1]
    grid = tuple(int(v) for v in G[f"solve/{name}/grid"])
    ptr, idx, val = orc.poisson3d(*grid)
    b = G[f"solve/{name}/b"]
    kw = {}
    if solver == "gmres":
        kw["restart"] = int(name.split("_r")[-1])
    x, it, rc, resid, rh = getattr(orc, solver)(ptr, idx, val, b, precon=precon, tol=1e-12,
                                                maxiter=1000, **kw)
    assert [it, rc] == list(G[f"solve/{name}/iter_status"])
    assert resid == G[f"solve/{name}/resid"][0]
    assert np.array_equal(x, G[f"solve/{name}/x"])
    assert np.array_equal(rh[1:it + 1], G[f"solve/{name}/rhistory"][1:it + 1])


def test_known_iteration_counts_32cubed():
    ptr, idx, val = orc.poisson3d(32, 32, 32)
    b = orc.spmv_csr(ptr, idx, val, np.ones(32 ** 3))
    _, it, rc, resid, _ = orc.cg(ptr, idx, val, b, precon="jacobi", maxiter=1000)
    assert [it, resid] == list(G["known/cg_jacobi_32/iter_resid"])          # 103
    _, it, rc, resid, _ = orc.bicgstab(ptr, idx, val, b, maxiter=1000)
    assert [it, resid] == list(G["known/bicgstab_none_32/iter_resid"])      # 75 at 1 thread
    _, it, rc, resid, _ = orc.gmres(ptr, idx, val, b, maxiter=1000, restart=30)
    assert [it, resid] == list(G["known/gmres30_none_32/iter_resid"])       # 276


# ---------------------------------------------------------------- BASELINE config 4's class (irregular, long rows)
import hashlib  # noqa: E402
import json     # noqa: E402

IRR = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "irregular_golden.json")))


def _irr_matrix(name):
    return orc.fem3(22)[:3] if name == "fem3_22" else orc.unstructured_mesh(60000) if name == "mesh_60k" else orc.heavy_tail(30000)


@pytest.mark.parametrize("name", ["fem3_22", "tail", "mesh_60k"])
def test_irregular_fixture_oracle_product_and_counts(name):
    """the generators still produce the matrices the fixture was made from, and the oracle reproduces what the reference
    returned for them: the bits of y = A*x and -- same arithmetic at one thread -- the exact iteration counts"""
    ptr, idx, val = _irr_matrix(name)
    g = IRR[name]
    h = hashlib.sha256()
    for a in (ptr.astype(np.int32), idx.astype(np.int32), val):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == g["sha256"], "orc.fem3 / orc.heavy_tail drifted from tests/golden/irregular_golden.json"
    n = len(ptr) - 1
    x = np.cos(np.arange(n) * 0.01) + 1.25
    y = orc.spmv_csr(ptr, idx, val, x)
    assert hashlib.sha256(y.tobytes()).hexdigest() == g["y_sha256"]
    for opts, want in g["solves"].items():
        if want["iter"] > 200:
            continue                                      # the 1225-iteration GMRES run stays with the generator script
        tok = opts.split()
        solver, precon = tok[1], tok[tok.index("-p") + 1]
        kw = {"restart": 30} if solver == "gmres" else {}
        _, it, rc, resid, rh = getattr(orc, solver)(ptr, idx, val, y, precon=precon, tol=1e-12, maxiter=2000, **kw)
        assert (it, rc) == (want["iter"], want["status"]), opts
        assert resid == want["resid"], opts
        assert list(rh[:6]) == want["rhistory_head"], opts
