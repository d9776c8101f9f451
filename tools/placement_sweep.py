"""Does the placement of the streams in HBM matter?  The 512^3 products and a few vector passes with x / y / the values shifted by so many
bytes inside one allocation each (same physical pages for every point of the sweep): python tools/placement_sweep.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402

lib = lis_amd.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
SLACK = 8 << 20
dptr, didx, dval0 = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval0.ptr, None))
x0 = DA(n, np.float64)
chunk = 1 << 24
for s in range(0, n, chunk):
    part = np.modf(np.arange(s, min(n, s + chunk), dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    check(lib.liship_memcpy_h2d(x0.ptr + 8 * s, part.ctypes.data, part.nbytes, None))
    check(lib.liship_device_synchronize())
VAL, X, Y, Z = DA(nnz + SLACK // 8, np.float64), DA(n + SLACK // 8, np.float64), DA(n + SLACK // 8, np.float64), DA(n + SLACK // 8, np.float64)
print("device addresses: VAL %#x  X %#x  Y %#x  Z %#x" % (VAL.ptr, X.ptr, Y.ptr, Z.ptr), flush=True)
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval0.ptr, None))
yref = DA(n, np.float64)
check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval0.ptr, x0.ptr, yref.ptr, None))
ref = yref.to_host().view(np.uint64)
work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(4, np.float64)


def place(xo, yo, vo, zo=0):
    check(lib.liship_memcpy_d2d(X.ptr + xo, x0.ptr, 8 * n, None))
    if vo is not None:
        check(lib.liship_memcpy_d2d(VAL.ptr + vo, dval0.ptr, 8 * nnz, None))
    check(lib.liship_memcpy_d2d(Z.ptr + zo, x0.ptr, 8 * n, None))
    check(lib.liship_device_synchronize())


def product(streamed, xo, yo, vo):
    lib.liship_spmv_csr_set_row_values(0 if streamed else 1)
    ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, VAL.ptr + vo, X.ptr + xo, Y.ptr + yo, None)), iters=30, warm=10)
    out = np.empty(n, np.uint64)
    check(lib.liship_memcpy_d2h(out.ctypes.data, Y.ptr + yo, 8 * n, None))
    check(lib.liship_device_synchronize())
    assert np.array_equal(out, ref)
    lib.liship_spmv_csr_set_row_values(1)
    return ms


OFFS = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304]
ODD = [0, 768, 1280, 4352, 12544, 36096, 69888, 135424, 200960, 331776, 593920]
last_v = None
for streamed in (1, 0):
    name = "values streamed (pattern7)" if streamed else "value records (dominant pattern)"
    print(f"## {name}", flush=True)
    base = None
    for which in ("Y", "X", "V") if streamed else ("Y", "X"):
        for off in OFFS + ODD[1:]:
            xo, yo, vo = (off if which == "X" else 0), (off if which == "Y" else 0), (off if which == "V" else 0)
            place(xo, yo, vo if (streamed and vo != last_v) else None)
            last_v = vo if streamed else last_v
            ms = product(streamed, xo, yo, vo)
            if off == 0 and base is None:
                base = ms
            print(f"  {which}_OFF {off:8d}: {ms:.4f} ms  ({ms / base:.3f} of the aligned placement)", flush=True)
# vector passes: y += a x (24 B per element), <x, y> (16 B), z = x + a y ... through the library's own kernels
print("## vector passes", flush=True)
for nm, call, nbytes in (("axpy  y += a x", lambda xo, yo: lib.liship_axpy_f64(n, 0.25, X.ptr + xo, Y.ptr + yo, None), 24),
                         ("dot   <x, y>", lambda xo, yo: lib.liship_dot_f64(n, X.ptr + xo, Y.ptr + yo, res.ptr, work.ptr, None), 16)):
    base = None
    for off in OFFS[:14] + ODD[1:6]:
        place(off, 0, None)
        ms = timed(lib, lambda: check(call(off, 0)), iters=50, warm=10)
        base = base or ms
        print(f"  {nm}  X_OFF {off:8d}: {ms:.4f} ms  {nbytes * n / ms / 1e6 / 8000:.3f} of 8 TB/s  ({ms / base:.3f})", flush=True)
