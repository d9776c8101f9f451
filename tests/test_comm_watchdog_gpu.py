"""First contact with several ranks must FAIL LOUDLY, never hang (the 8-GPU run is the driver's, not ours): the watchdogs of lis_comm.c / runtime.hip on the one GPU
this box has.
  * a stream that does not drain within the limit: liship_stream_synchronize returns LISHIP_ERR_TIMEOUT instead of blocking;
  * a communicator whose peer never joins: ncclCommInitRank runs on a helper thread, the caller aborts with a message when LIS_AMD_COMM_TIMEOUT expires;
  * two ranks that both sit on GPU 0 (RCCL refuses duplicates): both processes come back -- with an error code or the watchdog's abort -- and none hangs."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import lis_amd
from lis_amd import DeviceArray as DA, check

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_stream_watchdog_returns_instead_of_blocking():
    lib = lis_amd.load()
    assert lis_amd.gpu_available()
    n = 1 << 27                                              # 1 GiB vectors: an axpy is ~0.5 ms
    x, y = DA(n, np.float64), DA(n, np.float64)
    check(lib.liship_set_all_f64(n, 1.0, x.ptr, None)); check(lib.liship_set_all_f64(n, 0.0, y.ptr, None))
    check(lib.liship_device_synchronize())
    try:
        check(lib.liship_set_sync_timeout(0.02))
        for _ in range(400):                                 # ~0.2 s of queued work against a 20 ms limit
            check(lib.liship_axpy_f64(n, 1.0, x.ptr, y.ptr, None))
        t0 = time.time()
        rc = lib.liship_stream_synchronize(None)
        waited = time.time() - t0
        assert rc == -2 and 0.015 <= waited < 0.15, (rc, waited)       # LISHIP_ERR_TIMEOUT, and it came back at the limit
        check(lib.liship_set_sync_timeout(30.0))
        assert lib.liship_stream_synchronize(None) == 0      # within a generous limit the same wait succeeds ...
    finally:
        check(lib.liship_set_sync_timeout(0.0))
    check(lib.liship_device_synchronize())
    res, work = DA.zeros(2, np.float64), DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    check(lib.liship_nrm1_f64(8, y.ptr, res.ptr, work.ptr, None))
    assert res.to_host()[0] == 8 * 400.0                     # ... and every launch did run


def _spawn(rank, world, uidfile, timeout_s):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LIS_AMD_COMM_TIMEOUT=str(timeout_s))
    return subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_dup_worker.py"), str(rank), str(world), uidfile],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def test_a_peer_that_never_joins_aborts_the_rank_with_a_message(tmp_path):
    p = _spawn(0, 2, str(tmp_path / "uid"), 5)
    out, err = p.communicate(timeout=120)                    # (the limit is 5 s: far inside this)
    assert p.returncode == -6, (p.returncode, out[-500:], err[-1500:])          # SIGABRT from the watchdog
    assert "ncclCommInitRank did not return within 5 s" in err and "aborting" in err
    assert "RESULT" not in out


def test_two_ranks_on_one_gpu_come_back(tmp_path):
    uidfile = str(tmp_path / "uid")
    procs = [_spawn(r, 2, uidfile, 40) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=200) + (p.returncode,))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung in lis_amd_comm_init_rccl although LIS_AMD_COMM_TIMEOUT was set")
    for out, err, code in outs:
        came_back_with_error = "RESULT" in out and "rc=0 " not in out
        aborted_by_watchdog = code == -6 and "aborting" in err
        formed = "RESULT" in out and "rc=0 " in out         # (a RCCL build that accepts two ranks on one GPU: fine too)
        assert came_back_with_error or aborted_by_watchdog or formed, (code, out[-500:], err[-1500:])
