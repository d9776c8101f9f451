"""The native CSR / ELL / DIA / BSR products of the G^3 7-point matrix in the reference layout mode, `iters` launches each -- the process rocprofv3 profiles for
profiles/r06_formats_512_*, and a quick A/B timer by itself:   python tools/formats_child.py G iters [fmt,fmt,...] [--default]
Prints HIP-event ms per launch and the fraction of 8 TB/s on SURVEY 8d's bytes (CSR 12 nnz + 20 n, ELL 100 n, DIA 72 n; BSR 2x2: its stored bytes)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import lis_amd  # noqa: E402
import lisdrv  # noqa: E402
from lis_amd import _capi as capi, check  # noqa: E402

G, iters = int(sys.argv[1]), int(sys.argv[2])
fmts = sys.argv[3].split(",") if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else ["csr", "ell", "dia", "bsr"]
default = "--default" in sys.argv
plane = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--plane=")]
lib = lis_amd.load()
dll = lib.dll
assert lib.initialize([]) == 0
if plane:
    lib.liship_spmv_formats_set_plane(plane[0])
dll.lis_amd_set_residency(1)
if not default:
    assert dll.lis_amd_set_reference_layout(1) == 0
n, nnz = G ** 3, 7 * G ** 3 - 6 * G * G
A = capi.PM()
assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0 and lib.lis_matrix_set_size(A, 0, n) == 0
dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
assert dll.lis_amd_matrix_poisson3d(A, G, G, G, 1) == 0
x, y = capi.PV(), capi.PV()
for v in (x, y):
    assert lib.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(v)) == 0
for s0 in range(0, n, 1 << 24):
    cnt = min(1 << 24, n - s0)
    part = np.modf(np.arange(s0, s0 + cnt, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, s0, cnt, part.ctypes.data_as(capi.P_DBL), x) == 0
dll.lis_amd_stream.restype = C.c_void_p
stream = dll.lis_amd_stream()
timer, ev = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))
for fmt in fmts:
    M = A if fmt == "csr" else lisdrv.convert(lib, A, fmt)
    for _ in range(5):
        assert lib.lis_matvec(M, x, y) == 0
    dll.lis_amd_synchronize()
    check(lib.liship_timer_start(timer, stream))
    for _ in range(iters):
        assert lib.lis_matvec(M, x, y) == 0
    check(lib.liship_timer_stop(timer, stream))
    dll.lis_amd_synchronize()
    check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
    ms = ev.value / iters
    if fmt == "csr":
        B = 12 * nnz + 20 * n + 4
    elif fmt == "ell":
        B = 12 * M.contents.maxnzr * n + 16 * n
    elif fmt == "dia":
        B = 8 * M.contents.nnd * n + 16 * n
    else:
        B = 8 * M.contents.bnnz * 4 + 4 * M.contents.bnnz + 4 * (M.contents.nr + 1) + 16 * n
    print(f"{fmt} {G}^3: {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s  bytes {B}  frac {B / (ms * 1e-3) / 8e12:.4f}", flush=True)
    if M is not A:
        lib.lis_matrix_destroy(M)
