"""Short rows through the block-local kernel (round 6): a 7-point matrix with varying coefficients on a G^3 grid, kernel level, NO column codes -- the row-gather kernel on the 4 B
indices, the block-local kernel on the natural numbering, and the block-local kernel on the plan's renumbered form (compact cells).   python tools/local_short_rows_probe.py [G=160]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import lis_amd  # noqa: E402
import orc  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "mesh" else 160
lib = lis_amd.load()
if len(sys.argv) > 2 and sys.argv[1] == "mesh":            # python tools/local_short_rows_probe.py mesh NODES: the unstructured mesh of tests/orc.py (4096 Morton cells, random order inside a cell)
    cells = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    ptr, idx, val = orc.unstructured_mesh(int(sys.argv[2]), cells=cells)
    G = f"mesh {sys.argv[2]} cells {cells}^3"
else:
    ptr, idx, val = orc.poisson3d(G, G, G)
    val = val * np.random.default_rng(8).uniform(0.5, 1.5, len(val))
n, nnz = len(ptr) - 1, len(idx)
x = np.modf(np.arange(n) * 0.6180339887498949)[0] - 0.5
yref = orc.spmv_csr(ptr, idx, val, x)
dptr, didx, dval, dx = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64), DA.from_host(x, np.float64)
timer, ms = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))


def timed(plan, p, i, v, xx, reps=50):
    dy = DA.from_host(np.full(n, np.nan), np.float64)
    for _ in range(5):
        check(lib.liship_spmv_csr_f64(plan, p, i, v, xx, dy.ptr, None))
    check(lib.liship_timer_start(timer, None))
    for _ in range(reps):
        check(lib.liship_spmv_csr_f64(plan, p, i, v, xx, dy.ptr, None))
    check(lib.liship_timer_stop(timer, None)); check(lib.liship_device_synchronize()); check(lib.liship_timer_elapsed_ms(timer, C.byref(ms)))
    return ms.value / reps, dy


B = 12 * nnz + 20 * n + 4
for short in (0, 1):
    lib.liship_spmv_csr_set_local_short_rows(short)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_localize_columns(plan, dptr.ptr, didx.ptr, None))
    listed = lib.liship_csr_plan_localized(plan)
    t, dy = timed(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr)
    assert np.array_equal(dy.to_host(), yref)
    print(f"G={G} short-row lists {'on ' if short else 'off'}: listed {listed:>9d}  {t:.4f} ms  {B / t / 1e6 / 80:.1f} % of 8 TB/s on contract bytes", flush=True)
    if short:
        check(lib.liship_csr_plan_reorder(plan, dptr.ptr, didx.ptr, dval.ptr, 0, None))
        re = lib.liship_csr_plan_reordered(plan)
        print(f"   renumbered form: listed {re}")
        if re:
            inner, rp, ri, rv, pm = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(lib.liship_csr_plan_reordered_form(plan, C.byref(inner), C.byref(rp), C.byref(ri), C.byref(rv), C.byref(pm)))
            xp = DA(n, np.float64)
            check(lib.liship_permute_gather_f64(n, pm, dx.ptr, xp.ptr, None))
            t2, dy2 = timed(inner, rp, ri, rv, xp.ptr)
            yb = DA.from_host(np.full(n, np.nan), np.float64)
            check(lib.liship_permute_scatter_f64(n, pm, dy2.ptr, yb.ptr, None))
            assert np.array_equal(yb.to_host(), yref)
            print(f"   the product on P A P^T (what lis_solve iterates on): {t2:.4f} ms  {B / t2 / 1e6 / 80:.1f} %")
            lib.liship_spmv_csr_set_reorder(2)              # single products THROUGH the form: x gathered, rows stored at y[perm[r]] (opt-in: LIS_AMD_REORDER_PRODUCTS=1)
            t4, dy4 = timed(plan, dptr.ptr, didx.ptr, dval.ptr, dx.ptr)
            lib.liship_spmv_csr_set_reorder(1)
            assert np.array_equal(dy4.to_host(), yref)
            print(f"   a single product through the form (gather x + P A P^T + scattered y): {t4:.4f} ms against {t:.4f} in the caller's numbering")
    check(lib.liship_csr_plan_destroy(plan))
