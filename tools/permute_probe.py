"""liship_permute_gather_f64 / _scatter_f64 on a vector of the Queen class's size, the permutation made of scrambled triples; x evicted between launches by a 4 GB fill.
    python tools/permute_probe.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lis_amd
from lis_amd import DeviceArray as DA, check
lib = lis_amd.load()
n = 4102893
nodes = n // 3
rng = np.random.default_rng(1)
perm = (3 * rng.permutation(nodes)[:, None] + np.arange(3)[None, :]).reshape(-1).astype(np.int32)
dperm, x, xp = DA.from_host(perm, np.int32), DA.from_host(rng.uniform(-1, 1, n), np.float64), DA(n, np.float64)
big = DA(1 << 29, np.float64)          # 4 GB
timer, ev = C.c_void_p(), C.c_float()
check(lib.liship_timer_create(C.byref(timer)))
ref = x.to_host()[perm]
for mode in (1,):
    for evict in (0, 1):
        ts = []
        for rep in range(8):
            if evict:
                check(lib.liship_memset(big.ptr, rep, 8 << 29, None))
            check(lib.liship_timer_start(timer, None))
            check(lib.liship_permute_gather_f64(n, dperm.ptr, x.ptr, xp.ptr, None))
            check(lib.liship_timer_stop(timer, None))
            check(lib.liship_device_synchronize())
            check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
            ts.append(ev.value * 1e3)
        ok = np.array_equal(xp.to_host(), ref)
        print(f"gather mode {mode} evict={evict}: min {min(ts[2:]):.1f} us  median {sorted(ts[2:])[3]:.1f} us  ok={ok}", flush=True)
lib.liship_spmv_csr_set_reorder(1)
for evict in (0, 1):
    ts = []
    for rep in range(8):
        if evict:
            check(lib.liship_memset(big.ptr, rep, 8 << 29, None))
        check(lib.liship_timer_start(timer, None))
        check(lib.liship_permute_scatter_f64(n, dperm.ptr, xp.ptr, x.ptr, None))
        check(lib.liship_timer_stop(timer, None))
        check(lib.liship_device_synchronize())
        check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
        ts.append(ev.value * 1e3)
    print(f"scatter evict={evict}: min {min(ts[2:]):.1f} us  median {sorted(ts[2:])[3]:.1f} us", flush=True)
