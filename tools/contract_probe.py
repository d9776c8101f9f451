"""The contract form of the CSR product (4 B indices + 8 B values streamed: spmv_csr_rowgather_kernel) and the coded kernel at N^3, with the XCD strips on / off.

    python tools/contract_probe.py [N=512] [launches=30]

Plans: `bare` = liship_csr_plan_create + scan_band alone (what LIS_AMD_NO_INDEX_CODES=1 builds: geometry 192 / 1408), `full` = the plan lis_matvec builds, with the later
forms switched off at run time (geometry 256 / 2048).  HIP-event ms per launch, the contract's bytes (12 nnz + 20 n + 4) over it as a fraction of 8 TB/s, and y compared
bit for bit between every pair of runs."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402

lib = lis_amd.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = N ** 3
nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, dptr.ptr, didx.ptr, dval.ptr, None))
x, y, y0 = DA(n, np.float64), DA(n, np.float64), DA(n, np.float64)
chunk = 1 << 24
for s in range(0, n, chunk):
    part = np.modf(np.arange(s, min(n, s + chunk), dtype=np.float64) * 0.6180339887498949)[0] - 0.5
    check(lib.liship_memcpy_h2d(x.ptr + 8 * s, part.ctypes.data, part.nbytes, None))
    check(lib.liship_device_synchronize())
work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
timer, ev = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))
alg = 12 * nnz + 20 * n + 4

bare, full = C.c_void_p(), C.c_void_p()
check(lib.liship_csr_plan_create(C.byref(bare), n, dptr.ptr, None))
check(lib.liship_csr_plan_scan_band(bare, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_create(C.byref(full), n, dptr.ptr, None))
check(lib.liship_csr_plan_encode_indices(full, dptr.ptr, didx.ptr, None))
check(lib.liship_csr_plan_encode_row_patterns(full, dptr.ptr, None))
check(lib.liship_csr_plan_encode_row_values(full, dptr.ptr, dval.ptr, None))
print("strip rows: bare", lib.liship_csr_plan_strip_rows(bare), "full", lib.liship_csr_plan_strip_rows(full), flush=True)


def run(plan, fused):
    if fused:
        check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, x.ptr, 1, res.ptr, work.ptr, None))
    else:
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))


def measure(tag, plan, codes, strips, fused=False, bytes_=alg, variant=0):
    lib.liship_spmv_csr_set_variant(variant)
    lib.liship_spmv_csr_set_index_codes(codes)
    lib.liship_spmv_csr_set_row_patterns(0)
    lib.liship_spmv_csr_set_row_values(0)
    lib.liship_spmv_csr_set_xcd_strips(strips)
    check(lib.liship_memset(y.ptr, 0xff, 8 * n, None))
    for _ in range(10):
        run(plan, fused)
    check(lib.liship_timer_start(timer, None))
    for _ in range(launches):
        run(plan, fused)
    check(lib.liship_timer_stop(timer, None))
    check(lib.liship_device_synchronize())
    check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
    ms = ev.value / launches
    check(lib.liship_axpy_f64(n, -1.0, y0.ptr, y.ptr, None))
    check(lib.liship_nrm1_f64(n, y.ptr, res.ptr, work.ptr, None))
    diff = res.to_host()[0]
    lib.liship_spmv_csr_set_variant(0)
    print(f"{tag:46s} strips={strips} {ms:7.4f} ms  {2e-6 * nnz / ms:7.1f} GFLOP/s  frac(bytes)={bytes_ / ms / 1e6 / 8000:.4f}  |y - y0|_1 = {diff}", flush=True)
    return ms


# y0: the bare plan in the natural order
lib.liship_spmv_csr_set_xcd_strips(0)
lib.liship_spmv_csr_set_index_codes(0)
check(lib.liship_spmv_csr_f64(bare, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y0.ptr, None))
check(lib.liship_device_synchronize())
for rep in range(2):
    for strips in (0, 1):
        measure("rowgather, bare plan (192/1408)", bare, 0, strips)
        measure("rowgather, full plan (256/2048)", full, 0, strips)
        measure("rowgather + fused dots, bare plan", bare, 0, strips, fused=True)
        measure("rowgather + fused dots, full plan", full, 0, strips, fused=True)
        measure("coded kernel, full plan (9 B/nnz)", full, 1, strips, bytes_=9 * nnz + 20 * n + 4)
lib.liship_spmv_csr_set_index_codes(1)
lib.liship_spmv_csr_set_row_patterns(1)
lib.liship_spmv_csr_set_row_values(1)
lib.liship_spmv_csr_set_xcd_strips(1)
