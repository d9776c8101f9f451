"""Pins the ORDER of this library's reductions (a wavefront butterfly, the wavefronts of a workgroup in order, fixed tree folds over the
workgroups' partials): made ON AN MI355X by the library itself -- the reference adds left to right and cannot produce these bits
 -- so that a change of a reduction kernel that is meant to keep the order can prove it.  Iteration counts of the
Krylov solvers hang on these bits.

    python tests/golden/make_golden_reduction_bits.py      (needs a GPU; rewrites tests/golden/reduction_bits.json)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import lis_amd  # noqa: E402
import orc  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402

SIZES = [1, 2, 63, 64, 65, 255, 1000, 2049, 4096, 65536, 65537, (1 << 20) + 3, 5_000_001]


def bits(a):
    return [format(int(v), "016x") for v in np.atleast_1d(a).view(np.uint64)]


def measure(lib):
    out = {}
    work = DA(lib.liship_reduce_work_bytes() // 8, np.float64)
    for n in SIZES:
        rng = np.random.default_rng(n)
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        dx, dy = DA.from_host(x, np.float64), DA.from_host(y, np.float64)
        res = DA.from_host(np.zeros(2), np.float64)
        check(lib.liship_dot_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
        out[f"dot/{n}"] = bits(res.to_host()[:1])
        check(lib.liship_sumsq_f64(n, dx.ptr, res.ptr, work.ptr, None))
        out[f"sumsq/{n}"] = bits(res.to_host()[:1])
        check(lib.liship_dot2_f64(n, dx.ptr, dy.ptr, res.ptr, work.ptr, None))
        out[f"dot2/{n}"] = bits(res.to_host()[:2])
    # the reduction epilogue of the CSR product: one partial per row block, folded by the same tree
    for name, (ptr, idx, val) in (("p3d_40x33x29", orc.poisson3d(40, 33, 29)), ("rand_30000", orc.random_csr(30000, 9, seed=8)),
                                  ("fem3_12", orc.fem3(12)[:3])):
        n = len(ptr) - 1
        ncols = max(n, int(idx.max()) + 1)
        rng = np.random.default_rng(len(idx))
        x, w = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, n)
        dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
        dx, dw, dyy = DA.from_host(x, np.float64), DA.from_host(w, np.float64), DA(n, np.float64)
        plan = C.c_void_p()
        check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
        for sq in (0, 1):
            res = DA.from_host(np.zeros(2), np.float64)
            rc = lib.liship_spmv_csr_dot_f64(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr, dyy.ptr, dw.ptr, sq, res.ptr, work.ptr, None)
            assert rc == 0, rc
            out[f"spmv_dot/{name}/{sq}"] = bits(res.to_host()[:1 + sq])
        check(lib.liship_csr_plan_destroy(plan))
    return out


if __name__ == "__main__":
    lib = lis_amd.load()
    assert lis_amd.gpu_available()
    json.dump(measure(lib), open(os.path.join(HERE, "reduction_bits.json"), "w"), indent=0, sort_keys=True)
    print("wrote reduction_bits.json")
