"""Run by test_comm_watchdog_gpu.py, one process per rank: lis_amd_comm_init_rccl for a world of `world` ranks, every rank on device 0 (which RCCL refuses: one
GPU, one rank).  rank 0 writes the unique id to `uidfile`, the others wait for it.  Prints what happened; the point of the test is that something DOES happen --
a returned error or the watchdog's abort -- within the communication time limit."""
import ctypes as C
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import lis_amd  # noqa: E402

rank, world, uidfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lib = lis_amd.load()
assert lib.initialize([]) == 0
uid = (C.c_char * 128)()
if rank == 0:
    assert lib.dll.lis_amd_comm_get_unique_id(uid) == 0
    with open(uidfile + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.rename(uidfile + ".tmp", uidfile)
else:
    t0 = time.time()
    while not os.path.exists(uidfile):
        if time.time() - t0 > 120:
            sys.exit("no unique id from rank 0")
        time.sleep(0.05)
    uid = (C.c_char * 128).from_buffer_copy(open(uidfile, "rb").read())
t0 = time.time()
rc = lib.dll.lis_amd_comm_init_rccl(uid, rank, world, 0)
print(f"RESULT rank {rank} init_rccl rc={rc} after {time.time() - t0:.1f} s halo_comm={lib.dll.lis_amd_comm_halo_communicator()}", flush=True)
lib.dll.lis_amd_comm_finalize()
