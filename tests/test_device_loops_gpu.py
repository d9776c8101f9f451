"""Device-driven CG / BiCGSTAB / BiCG / GMRES(m) (scalars in HBM, iterations enqueued in batches, one read-back per batch) must leave
exactly what the host-scalar loops leave: iteration count, status, residual, every residual-history entry and
every bit of x (the unfused loops group their reductions differently and are compared by the golden-vector tests instead) --
whatever the position of the converged iteration inside a batch of 16, for every way a loop
ends (tolerance, maxiter, breakdown, BiCGSTAB's half step)."""
import ctypes as C

import numpy as np
import pytest

import lis_amd
import lisdrv
import orc

pytestmark = pytest.mark.gpu
DEVICE, HOST, UNFUSED = 0, 1, 2


@pytest.fixture(scope="module")
def lib():
    lib = lis_amd.load()
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(0)
    yield lib
    lib.dll.lis_amd_set_loop_mode(DEVICE)


def nonsym(seed):
    """the 7-point stencil with its off-diagonal entries perturbed independently: non-symmetric, diagonally dominant"""
    ptr, idx, val = orc.poisson3d(16, 13, 11)
    val = val.copy()
    rows = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    off = idx != rows
    val[off] *= np.random.default_rng(seed).uniform(0.2, 1.0, int(off.sum()))
    return ptr, idx, val


def run_modes(lib, ptr, idx, val, b, options, fmt="csr", x0=None, modes=(DEVICE, HOST)):
    outs = []
    for mode in modes:
        assert lib.dll.lis_amd_set_loop_mode(mode) == 0
        A = lisdrv.make_csr(lib, ptr, idx, val)
        B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt)
        outs.append(lisdrv.solve(lib, B, b, options, x0=x0))
        lib.lis_matrix_destroy(B)
    lib.dll.lis_amd_set_loop_mode(DEVICE)
    return outs


def same(outs):
    a = outs[0]
    for o in outs[1:]:
        assert (o["iter"], o["status"], o["err"]) == (a["iter"], a["status"], a["err"])
        assert o["resid"] == a["resid"] or (np.isnan(o["resid"]) and np.isnan(a["resid"]))
        assert np.array_equal(o["x"], a["x"], equal_nan=True)
        assert np.array_equal(o["rhistory"], a["rhistory"], equal_nan=True)
    return a


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "bicg", "gmres -restart 7", "gmres -restart 30"])
@pytest.mark.parametrize("precon", ["none", "jacobi"])
@pytest.mark.parametrize("fmt", ["csr", "ell", "dia", "jad", "bsr", "csc"])
def test_bits_match_host_loops(lib, solver, precon, fmt):
    ptr, idx, val = orc.poisson3d(17, 12, 10, sort_cols=(fmt == "dia"))
    n = len(ptr) - 1
    b = np.random.default_rng(3).uniform(-1, 1, n)
    a = same(run_modes(lib, ptr, idx, val, b, f"-i {solver} -p {precon} -tol 1e-12 -maxiter 400 -print mem", fmt))
    assert a["status"] == 0 and a["iter"] > 16


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "bicg", "gmres -restart 6", "gmres"])
@pytest.mark.parametrize("maxiter", [0, 1, 5, 15, 16, 17, 31, 32, 33])
def test_maxiter_at_every_batch_position(lib, solver, maxiter):
    ptr, idx, val = orc.poisson3d(14, 13, 9)
    b = np.random.default_rng(4).uniform(-1, 1, len(ptr) - 1)
    a = same(run_modes(lib, ptr, idx, val, b, f"-i {solver} -p jacobi -tol 1e-14 -maxiter {maxiter} -print mem"))
    assert a["status"] != 0 and a["iter"] == maxiter + 1


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "bicg", "gmres -restart 5", "gmres -restart 40"])
@pytest.mark.parametrize("tol", ["1e-1", "1e-2", "1e-3", "1e-4", "1e-5", "1e-6", "1e-7", "1e-8", "1e-9", "1e-10", "1e-11"])
def test_converged_iteration_anywhere_in_a_batch(lib, solver, tol):
    ptr, idx, val = orc.poisson3d(20, 9, 8) if solver == "cg" else nonsym(11)
    b = np.random.default_rng(5).uniform(-1, 1, len(ptr) - 1)
    x0 = np.random.default_rng(6).uniform(-1, 1, len(ptr) - 1)
    a = same(run_modes(lib, ptr, idx, val, b, f"-i {solver} -p none -tol {tol} -maxiter 500 -print mem -initx_zeros false",
                       x0=x0))
    assert a["status"] == 0


@pytest.mark.parametrize("cond", ["nrm2_r", "nrm2_b", "nrm1_b"])
def test_convergence_conditions(lib, cond):
    ptr, idx, val = nonsym(2)
    b = np.random.default_rng(7).uniform(-1, 1, len(ptr) - 1)
    for solver in ("cg", "bicgstab", "bicg", "gmres -restart 9"):
        if solver == "cg":
            p, i, v = orc.poisson3d(11, 10, 9)
            bb = b[: len(p) - 1]
        else:
            p, i, v, bb = ptr, idx, val, b
        same(run_modes(lib, p, i, v, bb, f"-i {solver} -p jacobi -tol 1e-10 -conv_cond {cond} -print mem"))


def test_bicgstab_half_step_exit(lib):
    """A = 2 I: s = r - alpha v vanishes in the first iteration, the loop leaves at :240-258 after x += alpha*phat."""
    n = 4096
    ptr = np.arange(n + 1, dtype=np.int32)
    idx = np.arange(n, dtype=np.int32)
    val = np.full(n, 2.0)
    b = np.random.default_rng(8).uniform(-1, 1, n)
    for precon in ("none", "jacobi"):
        a = same(run_modes(lib, ptr, idx, val, b, f"-i bicgstab -p {precon} -tol 1e-12 -print mem"))
        assert a["iter"] == 1 and a["status"] == 0
        assert np.allclose(a["x"], b * 0.5, rtol=1e-14, atol=0)


def test_breakdowns(lib):
    n = 512
    ptr = np.arange(n + 1, dtype=np.int32)
    idx = np.arange(n, dtype=np.int32)
    b = np.ones(n)
    # CG on A = 0: <p,q> == 0 in the first iteration (lis_solver_cg.c:196-202)
    a = same(run_modes(lib, ptr, idx, np.zeros(n), b, "-i cg -p none -print mem"))
    assert a["status"] != 0 and a["iter"] == 1
    a = same(run_modes(lib, ptr, idx, np.zeros(n), b, "-i bicg -p none -print mem"))     # <p~,q> == 0 (lis_solver_bicg.c:228-236)
    assert a["status"] != 0 and a["iter"] == 1
    # BiCGSTAB on a rotation-like operator: <rtld, r> hits zero (:190-196) or omega does
    sw = idx.reshape(-1, 2)[:, ::-1].reshape(-1).astype(np.int32)           # swaps neighbours: A^2 = I, <r, A r> small
    val = np.ones(n)
    val[1::2] = -1.0
    bb = np.zeros(n)
    bb[::2] = 1.0
    same(run_modes(lib, ptr, sw, val, bb, "-i bicgstab -p none -maxiter 40 -print mem"))


def test_print_out_lines_are_the_same(lib, capfd):
    ptr, idx, val = orc.poisson3d(8, 7, 6)
    b = np.random.default_rng(9).uniform(-1, 1, len(ptr) - 1)
    texts = []
    libc = C.CDLL(None)
    libc.fflush(None)
    capfd.readouterr()
    for mode in (DEVICE, HOST):
        run_modes(lib, ptr, idx, val, b, "-i cg -p jacobi -tol 1e-10 -print all", modes=(mode,))
        libc.fflush(None)                               # the library prints through C stdio
        texts.append(capfd.readouterr().out)
    assert texts[0] == texts[1] and "relative residual" in texts[0]


def test_cg_jacobi_uniform_and_varying_diagonal(lib):
    """CG + Jacobi in the device-driven loop takes 1/diag as ONE double when the diagonal is constant (the 7-point stencil: 6) and
    reads the array when it is not; either way the host-scalar loop, which always reads the array, must leave the same bits"""
    ptr, idx, val = orc.poisson3d(15, 14, 9)
    n = len(ptr) - 1
    b = np.random.default_rng(8).uniform(-1, 1, n)
    a = same(run_modes(lib, ptr, idx, val, b, "-i cg -p jacobi -tol 1e-12 -maxiter 400 -print mem"))
    assert a["status"] == 0
    val = val.copy()
    rows = np.repeat(np.arange(n), np.diff(ptr))
    val[idx == rows] += np.random.default_rng(9).uniform(0.0, 2.0, n)          # still symmetric positive definite
    a = same(run_modes(lib, ptr, idx, val, b, "-i cg -p jacobi -tol 1e-12 -maxiter 400 -print mem"))
    assert a["status"] == 0
    val[idx == rows] = 6.0
    val[ptr[n // 2] + int(np.flatnonzero(idx[ptr[n // 2]:ptr[n // 2 + 1]] == n // 2)[0])] = np.nextafter(6.0, 7.0)   # one ulp in one row
    a = same(run_modes(lib, ptr, idx, val, b, "-i cg -p jacobi -tol 1e-12 -maxiter 400 -print mem"))
    assert a["status"] == 0


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "bicg"])
@pytest.mark.parametrize("precon", ["none", "jacobi"])
@pytest.mark.parametrize("fmt", ["csr", "ell", "jad"])
def test_graph_replay_leaves_the_same_bits(lib, solver, precon, fmt):
    """LIS_AMD_GRAPHS=1: from the second batch of 16 iterations on, a single-rank solve replays a hipGraph of one batch
    (lis_solver.c: dev_loop_run): same kernels, same arguments, same order -- every bit as with plain launches,
    wherever in a replayed batch the loop ends"""
    ptr, idx, val = orc.poisson3d(17, 12, 10)
    n = len(ptr) - 1
    b = np.random.default_rng(5).uniform(-1, 1, n)
    outs, replays = [], []
    for mode in (0, 1):                                        # plain launches (default), graph replay
        assert lib.dll.lis_amd_set_graphs(mode) == 0
        A = lisdrv.make_csr(lib, ptr, idx, val)
        B = A if fmt == "csr" else lisdrv.convert(lib, A, fmt)
        outs.append(lisdrv.solve(lib, B, b, f"-i {solver} -p {precon} -tol 1e-12 -maxiter 400 -print mem"))
        replays.append(lib.dll.lis_amd_last_solve_graph_replays())
        lib.lis_matrix_destroy(B)
    lib.dll.lis_amd_set_graphs(0)
    a = same(outs)
    assert a["status"] == 0 and a["iter"] > 32
    assert replays[0] == 0 and replays[1] == (a["iter"] - 1) // 16, replays
    lib.dll.lis_amd_set_graphs(0)


@pytest.mark.parametrize("maxiter", [17, 31, 32, 33, 47, 48, 49])
def test_graph_replay_at_every_batch_position(lib, maxiter):
    ptr, idx, val = orc.poisson3d(14, 13, 9)
    b = np.random.default_rng(6).uniform(-1, 1, len(ptr) - 1)
    outs = []
    for mode in (0, 1):
        lib.dll.lis_amd_set_graphs(mode)
        A = lisdrv.make_csr(lib, ptr, idx, val)
        outs.append(lisdrv.solve(lib, A, b, f"-i cg -p jacobi -tol 1e-14 -maxiter {maxiter} -print mem"))
        lib.lis_matrix_destroy(A)
        if mode == 1:                                          # only FULL batches after the first are replayed
            assert lib.dll.lis_amd_last_solve_graph_replays() == maxiter // 16 - 1
    lib.dll.lis_amd_set_graphs(0)
    a = same(outs)
    assert a["iter"] == maxiter + 1
