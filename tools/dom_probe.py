"""Value-record kernels A/B at N^3 Poisson (round 3: the dominant-pattern kernel): python tools/dom_probe.py [N] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402

lib = lis_amd.load()
if os.environ.get("DOM_MARCH"):         # A/B: 0 = the gathering dominant-pattern kernel instead of the z-marching one
    lib.liship_spmv_csr_set_dom_march(int(os.environ["DOM_MARCH"]))
if os.environ.get("NO_XCD_STRIPS") == "1":        # A/B: the 7-offset pattern kernel in the natural block order
    lib.liship_spmv_csr_set_xcd_strips(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
OFF = {k: int(os.environ.get(k + "_OFF", "0")) for k in "XYV"}          # A/B: shift x / y / the values by so many bytes (channel placement)


class Shifted:
    def __init__(self, count, dtype, off):
        self.buf = DA(count + off // np.dtype(dtype).itemsize + 1, dtype)
        self.ptr, self.count, self.dtype = self.buf.ptr + off, count, np.dtype(dtype)

    def to_host(self):
        out = np.empty(self.count, self.dtype)
        check(lib.liship_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None))
        check(lib.liship_device_synchronize())
        return out


dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), Shifted(nnz, np.float64, OFF["V"])
x, y, y2 = Shifted(n, np.float64, OFF["X"]), Shifted(n, np.float64, OFF["Y"]), DA(n, np.float64)
print("device addresses: val %#x  x %#x  y %#x  idx %#x" % (dval.ptr, x.ptr, y.ptr, didx.ptr), flush=True)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
chunk = 1 << 24
for s in range(0, n, chunk):
    part = np.modf(np.arange(s, min(n, s + chunk), dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    check(lib.liship_memcpy_h2d(x.ptr + 8 * s, part.ctypes.data, part.nbytes, None))
    check(lib.liship_device_synchronize())
plan = C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
print("row patterns:", lib.liship_csr_plan_row_patterns(plan), "value records:", lib.liship_csr_plan_value_records(plan), flush=True)
FORMS = {"one row/lane": 0x20000000, "two rows/lane": 0x20004000, "dominant plain": 0x10000000, "default": 0}
if os.environ.get("DOM_FORMS"):
    FORMS = {k: v for k, v in FORMS.items() if k in os.environ["DOM_FORMS"].split(",")}
lib.liship_spmv_csr_set_row_values(0)
lib.liship_spmv_csr_set_variant(0x20000000)
check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y2.ptr, None))
ref = y2.to_host().view(np.uint64)
for rep in range(reps):
    for name, var in (("values streamed", 0),):
        lib.liship_spmv_csr_set_variant(var)
        check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=30, warm=100 if rep == 0 else 10)
        same = bool(np.array_equal(y.to_host().view(np.uint64), ref))
        print(f"{name:16s} {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s  {(8.0 * nnz + 17.0 * n) / ms / 1e6 / 8000:.3f} of 8 TB/s on 8 B/nnz + 17 B/row  bit-identical: {same}", flush=True)
lib.liship_spmv_csr_set_variant(0)
lib.liship_spmv_csr_set_row_values(1)
for rep in range(reps):
    for name, var in FORMS.items():
        lib.liship_spmv_csr_set_variant(var)
        check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=50, warm=20)
        same = bool(np.array_equal(y.to_host().view(np.uint64), ref)) if rep == 0 else None
        print(f"{name:16s} {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s  {17e-9 * n / ms * 1e3 / 8000:.3f} of 8 TB/s on 17 B/row" + ("" if same is None else f"  bit-identical: {same}"), flush=True)
# the fused-dot forms (what the Krylov loops run): y = A x with <x, y> and ||y||^2 in the product's pass
work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
first = None
for rep in range(reps):
    for name, var in (("round-2 dot", 0x20000000), ("dom dot4 blocks", 0x4000), ("plain chunks dot", 0x10000000), ("tiles dot", 0)):
        lib.liship_spmv_csr_set_variant(var)
        call1 = lambda: check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 0, res.ptr, work.ptr, None))
        ms1 = timed(lib, call1, iters=50, warm=20)
        call = lambda: check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 1, res.ptr, work.ptr, None))
        ms = timed(lib, call, iters=50, warm=20)
        name = f"{name} ({ms1:.4f} one dot)"
        got = res.to_host().copy()
        if first is None or name == "plain chunks dot":
            first = got
        same = bool(np.array_equal(y.to_host().view(np.uint64), ref)) if rep == 0 else None
        print(f"{name:16s} {ms:.4f} ms  {2e-6 * nnz / ms:.1f} GFLOP/s  {25e-9 * n / ms * 1e3 / 8000:.3f} of 8 TB/s on 25 B/row  sums equal: {bool(np.array_equal(got, first))}" + ("" if same is None else f"  y bit-identical: {same}"), flush=True)
lib.liship_spmv_csr_set_variant(0)
# the same call with a guard flag installed (what the device-driven Krylov loops run): the flag is a scalar load of another kernel's store
if True:
    flag = DA.zeros(2, np.float64)
    lib.dll.liship_krylov_guard.argtypes = [C.c_void_p]
    lib.dll.liship_krylov_guard.restype = C.c_int
    for name, var in (("dom dot4 blocks", 0x4000), ("tiles dot", 0)):
        lib.liship_spmv_csr_set_variant(var)
        call1 = lambda: check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 0, res.ptr, work.ptr, None))
        base = timed(lib, call1, iters=50, warm=20)
        check(lib.dll.liship_krylov_guard(flag.ptr))
        ms1 = timed(lib, call1, iters=50, warm=20)
        check(lib.dll.liship_krylov_guard(None))
        print(f"{name}: one dot {base:.4f} ms, with a guard flag {ms1:.4f} ms", flush=True)
    lib.liship_spmv_csr_set_variant(0)
