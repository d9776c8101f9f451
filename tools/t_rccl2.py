"""Bring-up probe: can two ranks share ONE GPU under RCCL?  (It refuses -- "Duplicate GPU detected" -- which is why the
N>1 data path is brought up through the gloo callbacks: bench.py --comm callbacks, tests/test_distributed.py.)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/t_rccl2.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402
import lis_amd  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = lis_amd.load()
assert lib.initialize([]) == 0
uid = [None]
if rank == 0:
    buf = (C.c_char * 128)()
    assert lib.dll.lis_amd_comm_get_unique_id(buf) == 0
    uid[0] = bytes(buf)
dist.broadcast_object_list(uid, src=0)
rc = lib.dll.lis_amd_comm_init_rccl(uid[0], rank, world, 0)          # both ranks on device 0
print(f"rank {rank}: lis_amd_comm_init_rccl on a shared device -> {rc}", flush=True)
