import ctypes as C, os, sys, time
import numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import lis_amd, lisdrv, orc
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
for N in (10, 64, 128, 200):
    ptr, idx, val = orc.poisson3d(N, N, N)
    n = len(ptr) - 1
    A = lisdrv.make_csr(lib, ptr, idx, val)
    vx, vy = lisdrv.new_vector(lib, A, np.ones(n)), lisdrv.new_vector(lib, A)
    for _ in range(20): lib.lis_matvec(A, vx, vy)
    lib.dll.lis_amd_synchronize()
    t0 = time.perf_counter()
    reps = 2000
    for _ in range(reps): lib.lis_matvec(A, vx, vy)
    t1 = time.perf_counter()
    lib.dll.lis_amd_synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: enqueue {1e6*(t1-t0)/reps:.1f} us/call, with drain {1e6*(t2-t0)/reps:.1f} us/call", flush=True)
