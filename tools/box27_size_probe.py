"""27-point box stencil (constant coefficients) through the plan, marching on / off, per grid size: python tools/box27_size_probe.py N [N ...]"""
import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import lis_amd
from lis_amd import DeviceArray as DA, check
from spmv_sweep import timed
exec(open(os.path.join(ROOT, "tools", "stencil27_probe.py")).read().split("lib = lis_amd.load()")[0].split('"""', 2)[2])
lib = lis_amd.load()
for N in [int(a) for a in sys.argv[1:]] or [192]:
    ptr, idx, val = stencil27(N)
    n, nnz = len(ptr) - 1, len(idx)
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    x, y = DA.from_host(np.cos(0.01 * np.arange(n)) + 1.25, np.float64), DA(n, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    out = {}
    for rep in range(2):
        for m in (0, 1):
            lib.liship_spmv_csr_set_dom_march(m)
            ms = timed(lib, lambda: check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None)), iters=50, warm=20)
            out.setdefault(m, []).append((round(ms, 4), lib.liship_csr_plan_box27(plan)))
    lib.liship_spmv_csr_set_dom_march(1)
    print(f"N={N} n={n}: marching off {out[0]}  on {out[1]}  (ms, box27 flag); 16 B per row at the on-time: {16e-9 * n / out[1][-1][0] * 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    check(lib.liship_csr_plan_destroy(plan))
