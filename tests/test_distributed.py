"""The N>1 path with world_size 2 and 3 over gloo: partition, ghost renumbering, halo tables, halo exchange and
cross-rank folds (tests/dist_worker.py).  `host` mode runs anywhere; `device` mode runs the same ranks on
one GPU (each rank drives its HBM slice; the halo takes the host-callback route instead of RCCL)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(mode, world, timeout=600, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   LIS_AMD_DEVICE=str(rank) if mode == "rccl" else "0", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank}/{world} {mode} OK" in out, f"rank {rank}:\n{out[-3000:]}"


@pytest.mark.parametrize("world", [2, 3])
def test_partition_tables_and_halo_on_cpu(world):
    _launch("host", world)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_spmv_and_solvers_on_one_gpu(world):
    _launch("device", world)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"LIS_AMD_TEST_HALO_DELAY_MS": "40"}, {"LIS_AMD_TEST_HALO_DELAY_MS": "40", "LIS_AMD_NO_DIRECT_HALO": "1"},
                                 {"LIS_AMD_NO_DIRECT_HALO": "1"}, {"LIS_AMD_NO_OVERLAP": "1"}])
def test_late_halo_changes_nothing(env):
    """Two ranks on one GPU with the halo transport DELAYED (the exchange callback sleeps 40 / 80 ms before it moves anything, the
    ranks by different amounts): the interior rows of every product have long finished when the ghosts land, the boundary rows are
    queued behind the landing on the same stream, and every check of the distributed worker -- product slices bit-equal to the
    single-process product, iteration counts, residuals -- must come out as without the delay.  Also with the boundary planes packed
    instead of sent straight from x (LIS_AMD_NO_DIRECT_HALO=1), and with the overlap off."""
    _launch("device", 2, extra_env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_renumbered_solves_on_several_ranks(world):
    """every rank renumbers its own rows and owned columns inside its plan (ghost columns and halo slots keep their place, the export lists move with the rows):
    products keep the single-process bits, lis_solve iterates in the ranks' numberings (tests/dist_worker.py renumber_checks)"""
    _launch("renumber", world, timeout=900)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_distributed_over_rccl_one_gpu_per_rank(world):
    """The real N > 1 data path: an RCCL communicator with one GPU per rank (grouped ncclSend/ncclRecv halos on the second stream,
    ncclAllGather + rank-order folds, device-driven loops across ranks).  Every product slice must carry the bits of the
    single-process product, every solve the counts of the callback run.  Needs `world` GPUs: skipped on a one-GPU box."""
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs, {_gpu_count()} visible")
    _launch("rccl", world)
