"""Host-memory collectives for liblis_amd's callback communicator (lis_amd_comm_init_callbacks), carried by
torch.distributed (gloo).  Test / bring-up plumbing: the production data path is RCCL (lis_amd_comm_init_rccl).
Used by tests/dist_worker.py and by `bench.py --comm callbacks` (several ranks sharing one GPU, which RCCL refuses)."""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                       C.POINTER(C.c_double), C.POINTER(C.c_int))


class Callbacks(C.Structure):
    _fields_ = [("allgather", ALLGATHER), ("neighbor_exchange", EXCHANGE), ("ctx", C.c_void_p)]


def make_callbacks(world):
    def allgather(ctx, send, recv, nbytes):
        src = torch.from_numpy(np.frombuffer(C.string_at(send, nbytes), dtype=np.uint8).copy())
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, src)
        flat = torch.cat(outs).numpy()
        C.memmove(recv, flat.ctypes.data, nbytes * world)
        return 0

    delay = float(os.environ.get("LIS_AMD_TEST_HALO_DELAY_MS", "0")) * 1e-3      # tests: a halo that arrives late (odd ranks later still),
    late = [int(os.environ.get("LIS_AMD_TEST_HALO_DELAY_COUNT", "30"))]           # for the first so many exchanges of the process

    def exchange(ctx, nneib, neib, sendbuf, sptr, recvbuf, rptr):
        if delay > 0.0 and late[0] > 0:
            late[0] -= 1
            time.sleep(delay * (1 + dist.get_rank() % 2))
        reqs, recvs = [], []
        for i in range(nneib):
            sc, rc = sptr[i + 1] - sptr[i], rptr[i + 1] - rptr[i]
            if sc > 0:
                t = torch.from_numpy(np.ctypeslib.as_array(sendbuf, shape=(sptr[nneib],))[sptr[i]:sptr[i + 1]].copy())
                reqs.append(dist.isend(t, dst=neib[i]))
            if rc > 0:
                t = torch.empty(rc, dtype=torch.float64)
                recvs.append((t, rptr[i]))
                reqs.append(dist.irecv(t, src=neib[i]))
        for r in reqs:
            r.wait()
        for t, off in recvs:
            arr = t.numpy()
            C.memmove(C.addressof(recvbuf.contents) + 8 * off, arr.ctypes.data, 8 * arr.size)
        return 0

    cb = Callbacks(ALLGATHER(allgather), EXCHANGE(exchange), None)
    return cb
