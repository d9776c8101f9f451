"""Kernel-variant sweep for the CSR SpMV on one GPU (development tool, not the bench contract).

    python tools/spmv_sweep.py [N ...]        # cubic grids, default 256 512
Prints per variant: ms/launch (HIP events), algorithmic GB/s and % of the 8 TB/s HBM3E peak.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402


def timed(lib, fn, iters=30, warm=25):
    t = C.c_void_p()
    check(lib.liship_timer_create(C.byref(t)))
    for _ in range(warm):
        fn()
    check(lib.liship_timer_start(t, None))
    for _ in range(iters):
        fn()
    check(lib.liship_timer_stop(t, None))
    ms = C.c_float()
    check(lib.liship_timer_elapsed_ms(t, C.byref(ms)))
    lib.liship_timer_destroy(t)
    return ms.value / iters


def main():
    lib = lis_amd.load()
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512]
    for N in sizes:
        n = N ** 3
        nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
        dptr, didx, dval = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64)
        x, y = DA(n, np.float64), DA(n, np.float64)
        variants = [int(v, 0) for v in os.environ.get("SWEEP_VARIANTS", "0,1,2,3").split(",")]
        for sorted_ in [int(v) for v in os.environ.get("SWEEP_SORTED", "0").split(",")]:
            check(lib.liship_poisson3d_csr(N, N, N, 0, n, sorted_, dptr.ptr, didx.ptr, dval.ptr, None))
            check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
            bytes_alg = 12 * nnz + 20 * n + 4
            for variant in variants:
                lib.liship_spmv_csr_set_variant(variant)
                plan = C.c_void_p()
                check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
                ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)))
                gbs = bytes_alg / ms / 1e6
                print(f"N={N} sorted={sorted_} variant={variant:#06x}: {ms:.4f} ms  {2*nnz/ms/1e6:.1f} GFLOP/s  "
                      f"{gbs:.0f} GB/s  {gbs/80:.1f}% of 8 TB/s", flush=True)
                lib.liship_csr_plan_destroy(plan)
            lib.liship_spmv_csr_set_variant(0)
            plan = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
            fw, fr = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
            for sq in (0, 1):
                ms = timed(lib, lambda: check(lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr,
                                                                          x.ptr, sq, fr.ptr, fw.ptr, None)))
                print(f"N={N} sorted={sorted_} fused dot (sumsq={sq}): {ms:.4f} ms  ({(bytes_alg + 8 * n) / ms / 1e6:.0f} GB/s incl. w)", flush=True)
            lib.liship_csr_plan_destroy(plan)
        # streaming yardsticks on the same box: copy (16 B/elem) and dot (16 B/elem), axpy (24 B/elem)
        ms = timed(lib, lambda: check(lib.liship_memcpy_d2d(y.ptr, x.ptr, 8 * n, None)))
        print(f"N={N} d2d copy: {ms:.4f} ms {16*n/ms/1e6:.0f} GB/s")
        ms = timed(lib, lambda: check(lib.liship_axpy_f64(n, 0.5, x.ptr, y.ptr, None)))
        print(f"N={N} axpy: {ms:.4f} ms {24*n/ms/1e6:.0f} GB/s")
        work, res = DA.zeros(lib.liship_reduce_work_bytes() // 8, np.float64), DA.zeros(2, np.float64)
        ms = timed(lib, lambda: check(lib.liship_dot_f64(n, x.ptr, y.ptr, res.ptr, work.ptr, None)))
        print(f"N={N} dot: {ms:.4f} ms {16*n/ms/1e6:.0f} GB/s")
        ms = timed(lib, lambda: check(lib.liship_nrm2_f64(n, x.ptr, res.ptr, work.ptr, None)))
        print(f"N={N} nrm2: {ms:.4f} ms {8*n/ms/1e6:.0f} GB/s")
        for a in (dptr, didx, dval, x, y):
            a.free()


if __name__ == "__main__":
    main()
