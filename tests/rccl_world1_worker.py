"""Run by test_rccl_world1_gpu.py in a process of its own (a communicator is per process): the same solves with
and without an RCCL communicator of ONE rank.  With it, every reduction of the device-driven loops goes through
ncclAllGather + the rank-order fold in liship_krylov_step, and every host-scalar reduction through lisc_fold --
the code an 8-GPU job runs, minus the neighbours.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import lis_amd  # noqa: E402
import lisdrv  # noqa: E402
import orc  # noqa: E402


def main():
    with_comm = sys.argv[1] == "rccl"
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    if with_comm:
        uid = (C.c_char * 128)()
        assert lib.dll.lis_amd_comm_get_unique_id(uid) == 0
        assert lib.dll.lis_amd_comm_init_rccl(uid, 0, 1, 0) == 0
    ptr, idx, val = orc.poisson3d(19, 13, 11)
    n = len(ptr) - 1
    b = np.random.default_rng(1).uniform(-1, 1, n)
    out = {}
    for mode in (0, 1):
        lib.dll.lis_amd_set_loop_mode(mode)
        for opts in ("-i cg -p jacobi", "-i cg -p none", "-i bicgstab -p none", "-i bicgstab -p jacobi", "-i gmres -restart 20", "-i bicg"):
            A = lisdrv.make_csr(lib, ptr, idx, val)
            o = lisdrv.solve(lib, A, b, opts + " -tol 1e-11 -print mem")
            out[f"{mode}:{opts}"] = dict(iter=o["iter"], status=o["status"], resid=o["resid"].hex() if hasattr(o["resid"], "hex") else float(o["resid"]).hex(),
                                         x=o["x"].tobytes().hex(), rh=o["rhistory"].tobytes().hex())
            lib.lis_matrix_destroy(A)
    if with_comm:
        assert lib.dll.lis_amd_comm_finalize() == 0
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
