"""north_star's "bit-exact iteration counts", made testable: the reference-order reduction mode.

lis_amd_set_reference_reductions(T) makes every sum of the library the sum the reference's OpenMP build forms with T threads -- T
contiguous chunks by LIS_GET_ISIE, each left to right from 0.0, the partials added serially (src/vector/lis_vector_ops.c:88-107,
:241-259).  Everything else on the path already is the reference's arithmetic (products bit-exact per row, element-wise passes one
rounded multiply + one rounded add), so with the mode on a whole solve must reproduce oracle/_ref AT THAT THREAD COUNT in every bit:
iteration count, status, the complete residual history, the solution.  Fixtures: tests/golden/rhistory_bits.{json,npz}, written by
tests/golden/make_golden_rhistory_bits.py from oracle/_ref at T = 1 and T = 8 (the counts differ between the two: BiCGSTAB 32^3 needs
75 iterations at T = 1 and 72 at T = 8, GMRES(30) 64^3 846 and 847 -- the mode follows both).  SLACK is 0 here.

The tree mode (default) keeps its own tests (test_configs_gpu.py, test_lisapi_gpu.py) with their documented slack.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import lis_amd
import lisdrv
import orc
import queen_class
from lis_amd import DeviceArray as DA, check
from lis_amd import _capi as capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "rhistory_bits.json")))
BITS = np.load(os.path.join(HERE, "golden", "rhistory_bits.npz"))
COMMON = " " + META["common_options"]
KEYS = sorted(META["solves"])
CASES = sorted({k.split("|")[0] for k in KEYS})


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(0)
    yield lib
    lib.dll.lis_amd_set_reference_reductions(0)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _from_file(lib, path):
    A, b, x = capi.PM(), capi.PV(), capi.PV()
    assert lib.lis_matrix_create(0, C.byref(A)) == 0
    assert lib.lis_vector_create(0, C.byref(b)) == 0 and lib.lis_vector_create(0, C.byref(x)) == 0
    assert lib.lis_input(A, b, x, path.encode()) == 0
    lib.lis_vector_destroy(b)
    lib.lis_vector_destroy(x)
    return A


_cache = {}


def _case(lib, case):
    """-> (A, b): the matrix in this library, b = A*1 by this library's product (bit-exact with the reference's, test_golden / test_io)"""
    if case in _cache:
        return _cache[case]
    if case.startswith("p3d_"):                       # the -storage cases: lis_solve converts the caller's matrix for good, so every solve gets a fresh one
        l, m, n = (int(v) for v in case[4:].split("x"))
        ptr, idx, val = orc.poisson3d(l, m, n, sort_cols=True)
        A = lisdrv.make_csr(lib, ptr, idx, val)
        return A, orc.spmv_csr(ptr, idx, val, np.ones(len(ptr) - 1))
    if case.startswith("poisson"):
        N = int(case[len("poisson"):])
        ptr, idx, val = orc.poisson3d(N, N, N)
        A = lisdrv.make_csr(lib, ptr, idx, val)
    elif case.startswith("mm/"):
        A = _from_file(lib, os.path.join(HERE, "golden", case))
    else:
        path, _, _ = queen_class.generate("mini")
        try:
            A = _from_file(lib, path)
        finally:
            os.unlink(path)
    b = lisdrv.matvec(lib, A, np.ones(A.contents.n))
    _cache[case] = (A, b)
    return A, b


def _check(lib, key, loop_mode=0):
    case, opts, tag = key.split("|")
    T = int(tag[1:])
    want = META["solves"][key]
    A, b = _case(lib, case)
    assert len(b) == want["n"]
    assert lib.dll.lis_amd_set_reference_reductions(T) == 0
    lib.dll.lis_amd_set_loop_mode(loop_mode)
    try:
        res = lisdrv.solve(lib, A, b, opts + COMMON)
    finally:
        lib.dll.lis_amd_set_reference_reductions(0)
        lib.dll.lis_amd_set_loop_mode(0)
    assert (res["iter"], res["status"]) == (want["iter"], want["status"]), (key, res["iter"], res["status"])
    rh = BITS[key]
    assert len(res["rhistory"]) == len(rh)
    # every bit of every entry (view as integers: NaN-safe, -0.0-safe)
    diff = np.flatnonzero(res["rhistory"].view(np.int64) != rh.view(np.int64))
    assert diff.size == 0, (key, "first differing history entry", int(diff[0]), res["rhistory"][diff[0]].hex(), rh[diff[0]].hex())
    assert float(res["resid"]).hex() == want["resid_hex"]
    assert _sha(res["x"]) == want["x_sha256"], key
    if case.startswith("p3d_"):
        assert A.contents.matrix_type == capi.FORMAT_ID[opts.split()[-1]]      # converted for good, as in the reference (lis_solver.c:640-657)
        lib.lis_matrix_destroy(A)


@pytest.mark.parametrize("key", KEYS)
def test_solve_carries_the_reference_bits_at_its_thread_count(lib, key):
    """SLACK = 0: count, status, residual history (bits), final residual (bits), solution (sha256) of oracle/_ref at T threads."""
    _check(lib, key)


@pytest.mark.parametrize("key", [k for k in KEYS if k.startswith("poisson32|") or k.startswith("queen_mini|")])
@pytest.mark.parametrize("loop_mode", [1, 2])
def test_host_scalar_and_unfused_loops_carry_them_too(lib, key, loop_mode):
    """the same through the host-scalar loops (1) and the one-kernel-per-reference-call loops (2): the mode lives in the reductions, not in a loop"""
    _check(lib, key, loop_mode)


def _ref_sum(terms, T):
    """the reference's sum of `terms` with T threads: chunks by LIS_GET_ISIE, each left to right (np.add.accumulate is sequential), partials serially"""
    n = len(terms)
    total = np.float64(0.0)
    for t in range(T):
        if t < n % T:
            ie = n // T + 1
            is_ = ie * t
        else:
            ie = n // T
            is_ = ie * t + n % T
        ie += is_
        part = np.float64(0.0)
        if ie > is_:
            part = np.add.accumulate(np.concatenate(([0.0], terms[is_:ie])))[-1]
        total = total + part
    return total


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 100003])
@pytest.mark.parametrize("T", [1, 3, 8, 64])
def test_kernel_level_sums_are_the_chunked_left_to_right_sums(lib, n, T):
    check(lib.liship_set_device(0))
    rng = np.random.default_rng(n * 131 + T)
    x = rng.uniform(-1, 1, n) * 10.0 ** rng.integers(-8, 8, n)
    y = rng.uniform(-1, 1, n)
    dx, dy = DA.from_host(x), DA.from_host(y)
    res = DA.from_host(np.zeros(4))
    work = DA(lib.liship_reduce_work_bytes(), np.uint8)
    check(lib.liship_set_reference_reductions(T))
    try:
        assert lib.liship_get_reference_reductions() == T
        check(lib.liship_dot_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
        assert res.to_host()[0].hex() == float(_ref_sum(x * y, T)).hex()
        check(lib.liship_nrm2_f64(n, dx.ptr, res.ptr, work.ptr, None))
        assert res.to_host()[0].hex() == float(np.sqrt(_ref_sum(x * x, T))).hex()
        check(lib.liship_nrm1_f64(n, dx.ptr, res.ptr, work.ptr, None))
        assert res.to_host()[0].hex() == float(_ref_sum(np.abs(x), T)).hex()
        check(lib.liship_dot2_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
        got = res.to_host()
        assert (got[0].hex(), got[1].hex()) == (float(_ref_sum(x * y, T)).hex(), float(_ref_sum(x * x, T)).hex())
        # a fused pass: y += a*x ; {sum y^2, sum w*y} -- the element-wise part must be the tree mode's, the sums the ordered ones
        a = -0.37
        w = rng.uniform(-1, 1, n)
        dw, dy2 = DA.from_host(w), DA.from_host(y)
        check(lib.liship_axpy_sumsq_dot_f64(n, a, dx.ptr, dy2.ptr, dw.ptr, res.ptr, work.ptr, None))
        ynew = y + a * x
        assert np.array_equal(dy2.to_host(), ynew)
        got = res.to_host()
        assert (got[0].hex(), got[1].hex()) == (float(_ref_sum(ynew * ynew, T)).hex(), float(_ref_sum(w * ynew, T)).hex())
    finally:
        check(lib.liship_set_reference_reductions(0))
    # and off again: the tree's result (pinned elsewhere) differs from or equals the ordered one, but the switch must be off
    assert lib.liship_get_reference_reductions() == 0


def test_fused_dot_products_refuse_while_the_mode_is_on(lib):
    check(lib.liship_set_device(0))
    ptr, idx, val = orc.poisson3d(12, 10, 8)
    n = len(ptr) - 1
    x = np.random.default_rng(5).uniform(-1, 1, n)
    dptr, didx, dval, dx, dy, dw = (DA.from_host(a) for a in (ptr, idx, val, x, np.zeros(n), x))
    res = DA.from_host(np.zeros(4))
    work = DA(lib.liship_reduce_work_bytes(), np.uint8)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_set_reference_reductions(4))
    try:
        rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dy.ptr, dw.ptr, 1, res.ptr, work.ptr, None)
        assert rc == -1          # LISHIP_ERR_ARG: the caller runs the product and one ordered pass
    finally:
        check(lib.liship_set_reference_reductions(0))
        check(lib.liship_csr_plan_destroy(plan))


def test_environment_variable_selects_the_mode(tmp_path):
    """LIS_AMD_REFERENCE_REDUCTIONS=T is read by lis_initialize (a fresh process: the switch is applied where the device comes up)"""
    import subprocess
    import sys
    key = "poisson32|-i bicgstab -p none|T8"
    code = (
        "import sys, json, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import lis_amd, lisdrv, orc\n"
        "lib = lis_amd.load(); assert lib.initialize([]) == 0\n"
        "assert lib.dll.lis_amd_get_reference_reductions() == 8\n"
        "ptr, idx, val = orc.poisson3d(32, 32, 32)\n"
        "A = lisdrv.make_csr(lib, ptr, idx, val)\n"
        "b = lisdrv.matvec(lib, A, np.ones(len(ptr) - 1))\n"
        "res = lisdrv.solve(lib, A, b, %r)\n"
        "print('RESULT ' + json.dumps([int(res['iter']), [float(v).hex() for v in res['rhistory']]]), flush=True)\n"
    ) % (os.path.dirname(HERE), HERE, key.split("|")[1] + COMMON)
    env = dict(os.environ, LIS_AMD_REFERENCE_REDUCTIONS="8")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]      # (the C library's own stdout lines land around it)
    it, hist = json.loads(line[len("RESULT "):])
    assert it == META["solves"][key]["iter"]
    assert hist == [float(v).hex() for v in BITS[key]]
