import sys, os, ctypes as C
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch", torch.__version__, "cuda avail", torch.cuda.is_available())
import numpy as np
import lis_amd, __graft_entry__ as g
lib = lis_amd.load()
os.system(f"grep -E 'amdhip|rccl|hsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
g.smoke()
uid = (C.c_char * 128)()
rc = lib.dll.lis_amd_comm_get_unique_id(uid); print("get_unique_id", rc)
rc = lib.dll.lis_amd_comm_init_rccl(uid, 0, 1, 0); print("init_rccl", rc)
os.system(f"grep -E 'amdhip|rccl|hsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
if mode == "torch_first":
    x = torch.ones(10, device="cuda"); torch.cuda.synchronize(); print("torch cuda ok", float(x.sum()))
print("done", mode)
