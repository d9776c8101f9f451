"""Host cost of one lis_matvec call in resident mode (issue only) against the steady-state time per product.
3.3 us per call on the GPU box; at 200^3 the device needs 144 us per product, so the queue never runs dry.
    python tests/perf/call_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import lis_amd      # noqa: E402
import lisdrv       # noqa: E402
import orc          # noqa: E402

lib = lis_amd.load()
lib.initialize([])
lib.dll.lis_amd_set_residency(1)
for N in (16, 200):
    ptr, idx, val = orc.poisson3d(N, N, N)
    n = N ** 3
    A = lisdrv.make_csr(lib, ptr, idx, val)
    vx, vy = lisdrv.new_vector(lib, A, np.ones(n)), lisdrv.new_vector(lib, A)
    for _ in range(10):
        lib.lis_matvec(A, vx, vy)
    lib.dll.lis_amd_synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        lib.lis_matvec(A, vx, vy)
    t1 = time.perf_counter()
    lib.dll.lis_amd_synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: issue {1e6 * (t1 - t0) / 200:.1f} us/call, incl. drain {1e6 * (t2 - t0) / 200:.1f} us/call")
