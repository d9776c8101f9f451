"""per-call cost of lis_matvec in the three data policies (default COHERENT by page protection, eager COHERENT, RESIDENT): python tools/coherent_probe.py [N]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd, lisdrv, orc
from lis_amd import _capi as capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = lis_amd.load(); assert lib.initialize([]) == 0
dll = lib.dll
ptr, idx, val = orc.poisson3d(N, N, N)
n = N ** 3
for mode in ("default", "eager", "resident", "default"):
    dll.lis_amd_set_residency(1 if mode == "resident" else 0)
    dll.lis_amd_set_coherence(0 if mode == "eager" else 1)
    A = lisdrv.make_csr(lib, ptr, idx, val)
    x, y = lisdrv.new_vector(lib, A), lisdrv.new_vector(lib, A)
    for i in range(0, n, 1 << 20):
        cnt = min(1 << 20, n - i)
        ones = np.ones(cnt)
        lib.lis_vector_set_values2(0, i, cnt, ones.ctypes.data_as(capi.P_DBL), x)
    ts = []
    for it in range(30):
        t0 = time.perf_counter(); lib.lis_wtime()
        assert lib.lis_matvec(A, x, y) == 0
        lib.lis_wtime(); ts.append((time.perf_counter() - t0) * 1e3)
    nrm = C.c_double(); lib.lis_vector_nrm2(y, C.byref(nrm))
    print(f"{mode:9s} first {ts[0]:8.3f} ms  second {ts[1]:8.3f}  median of the rest {np.median(ts[2:]):8.4f} ms   2-norm {nrm.value:.6e}", flush=True)
    for v in (x, y): lib.lis_vector_destroy(v)
    lib.lis_matrix_destroy(A)
