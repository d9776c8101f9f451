"""PCIe-inclusive rate of a stand-alone lis_matvec in the default (coherent) residency mode: x goes up and y comes
down on every call.  DESIGN.md 1 quotes the number; it is never bench.py's `value`.   python tests/perf/coherent_rate.py"""
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
N = 256
n = N ** 3
ptr, idx, val = orc.poisson3d(N, N, N)
A = lisdrv.make_csr(lib, ptr, idx, val)
vx, vy = lisdrv.new_vector(lib, A, np.ones(n)), lisdrv.new_vector(lib, A)
lib.dll.lis_amd_set_residency(0)
for _ in range(3):
    lib.lis_matvec(A, vx, vy)
t = time.perf_counter()
for _ in range(10):
    lib.lis_matvec(A, vx, vy)
el = (time.perf_counter() - t) / 10
print(f"COHERENT lis_matvec 256^3: {el * 1e3:.2f} ms/call = {2 * len(idx) / el / 1e9:.1f} GFLOP/s "
      f"(x up + y down over PCIe: {2 * 8 * n / el / 1e9:.1f} GB/s)")
