"""Which CSR kernel for which mean row length?  Banded matrices with exactly L entries per row (columns r-L/2 ..),
~100 M non-zeros each, timed with the row-gather kernel (variant 0) and the products kernel (variants 0x4, 0x14).
Development tool behind the automatic choice in liship_csr_plan_create."""
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
    variants = [int(v, 0) for v in os.environ.get("SWEEP_VARIANTS", "0,0x4,0x14").split(",")]
    for L in [int(a) for a in sys.argv[1:]] or [4, 7, 10, 12, 16, 20, 24, 32, 48, 80]:
        n = int(100e6 // L)
        ptr = (np.arange(n + 1, dtype=np.int64) * L).astype(np.int32)
        cols = (np.arange(n, dtype=np.int64)[:, None] + (np.arange(L) - L // 2) * 3) % n       # stride-3 band, wraps around
        idx = cols.astype(np.int32).ravel()
        val = np.random.default_rng(L).uniform(-1, 1, n * L)
        dptr, didx, dval = DA.from_host(ptr), DA.from_host(idx), DA.from_host(val)
        x, y = DA.from_host(np.random.default_rng(1).uniform(-1, 1, n)), DA(n, np.float64)
        b = 12 * n * L + 20 * n
        out = []
        for v in variants:
            lib.liship_spmv_csr_set_variant(v)
            plan = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
            if os.environ.get("SWEEP_CODES"):
                check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
            if os.environ.get("SWEEP_LOCAL"):
                check(lib.liship_csr_plan_localize_columns(plan, dptr.ptr, didx.ptr, None))
            ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)))
            tag = ('c%d' % lib.liship_csr_plan_coded(plan) if lib.liship_csr_plan_coded(plan) else '') + \
                  ('L%.2f' % (lib.liship_csr_plan_localized(plan) / (n * L)) if lib.liship_csr_plan_localized(plan) else '')
            out.append(f"{v:#x}{tag}: {ms:.4f} ms {b / ms / 1e6:.0f} GB/s")
            lib.liship_csr_plan_destroy(plan)
        lib.liship_spmv_csr_set_variant(0)
        print(f"L={L} n={n}: " + " | ".join(out), flush=True)
        for d in (dptr, didx, dval, x, y):
            d.free()


if __name__ == "__main__":
    main()
