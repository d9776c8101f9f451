"""Short rows (7-point Poisson, N^3) under a numbering without locality: the grid's nodes permuted at random inside runs of BAND -- what the plan's kernels make of it.
    python tools/scrambled_short_rows_probe.py [N=160] [BAND=4096]"""
import ctypes as C, os, sys
import numpy as np
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import lis_amd, orc
from lis_amd import DeviceArray as DA, check
lib = lis_amd.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
band = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ptr, idx, val = orc.poisson3d(N, N, N)
n = len(ptr) - 1
rng = np.random.default_rng(5)
perm = np.arange(n)
for lo in range(0, n, band):
    perm[lo:lo + band] = lo + rng.permutation(min(band, n - lo))
inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
lens = np.diff(ptr)[perm]
ptr2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
src = np.repeat(np.asarray(ptr[:-1], np.int64)[perm] - ptr2[:-1], lens) + np.arange(ptr2[-1])
idx2, val2 = inv[idx[src]].astype(np.int32), val[src] * rng.uniform(0.5, 1.5, len(src))
timer, ev = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))
for tag, (p, i, v) in (("natural", (ptr, idx, val * rng.uniform(0.5, 1.5, len(val)))), ("scrambled", (ptr2, idx2, val2))):
    dptr, didx, dval = DA.from_host(p, np.int32), DA.from_host(i, np.int32), DA.from_host(v, np.float64)
    x, y = DA.from_host(rng.uniform(-1, 1, n), np.float64), DA(n, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_scan_band(plan, dptr.ptr, didx.ptr, None))
    for reorder in ((0, 1) if tag == "scrambled" else (0,)):
      if reorder:
        check(lib.liship_csr_plan_reorder(plan, dptr.ptr, didx.ptr, dval.ptr, 0, None))
        print("   reordered:", lib.liship_csr_plan_reordered(plan), "lines of x over the row blocks", flush=True)
      for _ in range(10):
        check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))
      check(lib.liship_timer_start(timer, None))
      for _ in range(30):
          check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, x.ptr, y.ptr, None))
      check(lib.liship_timer_stop(timer, None)); check(lib.liship_device_synchronize()); check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
      ms = ev.value / 30
      print(f"{tag:10s} n={n} coded={lib.liship_csr_plan_coded(plan)} patterns={lib.liship_csr_plan_row_patterns(plan)}  {ms:.4f} ms  {(12 * len(i) + 20 * n) / ms / 1e6 / 8000:.3f} of 8 TB/s on 12 B/nnz + 20 B/row", flush=True)

# the same numbering through the Lis API: lis_solve iterates in the plan's numbering (b, x0 gathered once, x scattered back)
if os.environ.get("SOLVES", "1") == "1":
    import lisdrv
    from lis_amd import _capi as capi
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(1)
    A = lisdrv.make_csr(lib, ptr2, idx2, val[src])          # (constant coefficients: the SPD matrix)
    b = orc.spmv_csr(ptr2, idx2, val[src], np.ones(n))
    for opts in ("-i cg -p jacobi", "-i bicgstab -p none", "-i gmres -restart 30 -p none"):
        for on in (1, 0):
            lib.liship_spmv_csr_set_reorder(on)
            out = lisdrv.solve(lib, A, b, opts + " -tol 1e-10 -maxiter 300")
            print(f"{opts:32s} renumbered={lib.dll.lis_amd_last_solve_renumbered()}  iter {out['iter']}  {out['iter'] / out['itime']:.0f} it/s  |x-1|max {np.abs(out['x'] - 1).max():.2e}", flush=True)
    lib.liship_spmv_csr_set_reorder(1)
