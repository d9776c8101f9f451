"""Does the relative placement of value[] / index[] / x / y in HBM matter?  One big allocation, the CSR arrays of the
512^3 stencil generated at controlled offsets inside it, the same kernel timed per placement.
    python tools/align_sweep.py [N]"""
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
    n = N ** 3
    nnz = lib.liship_poisson3d_nnz(N, N, N, 0, n)
    sizes = {"ptr": 4 * (n + 1), "idx": 4 * nnz, "val": 8 * nnz, "x": 8 * n, "y": 8 * n}
    slack = 64 << 20
    pool = DA(sum(sizes.values()) + 8 * slack, np.uint8)
    bytes_alg = 12 * nnz + 20 * n + 4
    rng = np.random.default_rng(1)
    results = []
    for trial in range(14):
        if trial == 0:
            offs = {k: 0 for k in sizes}
        elif trial < 8:
            offs = {k: int(rng.integers(0, 1 << 14)) * 4096 for k in sizes}          # random 4 KiB-granular shifts < 64 MiB
        else:
            offs = {k: int(rng.integers(0, 1 << 8)) * 256 for k in sizes}            # random 256 B-granular shifts < 64 KiB
        base, p = pool.ptr, {}
        for k in ("val", "idx", "ptr", "x", "y"):
            p[k] = base + offs[k]
            base += sizes[k] + slack
            base = (base + 4095) & ~4095
        check(lib.liship_poisson3d_csr(N, N, N, 0, n, 0, p["ptr"], p["idx"], p["val"], None))
        check(lib.liship_set_all_f64(n, 1.0, p["x"], None))
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, p["ptr"], None))
        ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, p["ptr"], p["idx"], p["val"], p["x"], p["y"], None)))
        lib.liship_csr_plan_destroy(plan)
        results.append(ms)
        print(f"trial {trial:2d}: {ms:.4f} ms  {bytes_alg / ms / 1e6 / 80:.1f}%  offsets(KiB) " +
              " ".join(f"{k}={offs[k] / 1024:.2f}" for k in ("val", "idx", "ptr", "x", "y")), flush=True)
    print(f"min {min(results):.4f} max {max(results):.4f} ms")


if __name__ == "__main__":
    main()
