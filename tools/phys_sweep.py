"""Does the PHYSICAL placement matter?  Several independent allocations of the 512^3 CSR arrays held at once
(so each lands on different physical pages), the same kernel timed on each.    python tools/phys_sweep.py [N] [copies]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402
from spmv_sweep import timed  # noqa: E402


def main():
    lib = lis_amd.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = N ** 3
    nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
    bytes_alg = 12 * nnz + 20 * n + 4
    keep = []
    for c in range(copies):
        ptr, idx, val, x, y = DA(n + 1, np.int32), DA(nnz, np.int32), DA(nnz, np.float64), DA(n, np.float64), DA(n, np.float64)
        keep.append((ptr, idx, val, x, y))
        check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, ptr.ptr, idx.ptr, val.ptr, None))
        check(lib.liship_set_all_f64(n, 1.0, x.ptr, None))
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, ptr.ptr, None))
        for rep in range(2):
            ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, ptr.ptr, idx.ptr, val.ptr, x.ptr, y.ptr, None)))
            print(f"copy {c} rep {rep}: {ms:.4f} ms  {bytes_alg / ms / 1e6 / 80:.1f}%  val@{val.ptr:#x}", flush=True)
        lib.liship_csr_plan_destroy(plan)


if __name__ == "__main__":
    main()
