"""BASELINE config 4's stand-in at Queen's scale (tests/golden/gen_queen_class.c) through lis_input, then timed products and solves:
    python tools/queen_probe.py [reps]        (env QUEEN_VARIANT: liship_spmv_csr_set_variant bits, QUEEN_NO_LOCAL=1: no block-local columns)"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd, lisdrv, queen_class
from lis_amd import _capi as capi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = lis_amd.load(); assert lib.initialize([]) == 0
dll = lib.dll
dll.lis_amd_set_residency(1)
if os.environ.get("QUEEN_NO_LOCAL") == "1":
    lib.liship_spmv_csr_set_local_columns(0)
if os.environ.get("QUEEN_ROUND3") == "1":              # round-3 form: 4096-item blocks, positions through LDS (two workgroups per CU on this matrix)
    lib.liship_spmv_csr_set_local_register_positions(0)
if os.environ.get("QUEEN_BAND"):                      # numbering sensitivity: 1 = the natural (lexicographic) node order, 1024 = the fixture's scramble
    queen_class.CASES["full"] = (queen_class.CASES["full"][0], int(os.environ["QUEEN_BAND"]))
t0 = time.time(); path, rows, stored = queen_class.generate("full"); t_gen = time.time() - t0
A, b, x0 = capi.PM(), capi.PV(), capi.PV()
lib.lis_matrix_create(0, C.byref(A)); lib.lis_vector_create(0, C.byref(b)); lib.lis_vector_create(0, C.byref(x0))
t0 = time.time(); assert lib.lis_input(A, b, x0, path.encode()) == 0; t_read = time.time() - t0
os.unlink(path)
n, nnz = A.contents.n, A.contents.nnz
dll.lis_amd_matrix_local_columns.argtypes = [capi.PM]
t0 = time.time(); listed = dll.lis_amd_matrix_local_columns(A); t_plan = time.time() - t0       # (the first call uploads the matrix and builds the plan)
xs = np.cos(np.arange(n) * 0.01) + 1.25
vx, vy = lisdrv.new_vector(lib, A, xs), lisdrv.new_vector(lib, A)
if os.environ.get("QUEEN_VARIANT"):
    lib.liship_spmv_csr_set_variant(int(os.environ["QUEEN_VARIANT"], 0))
def timed():
    for _ in range(10):
        assert lib.lis_matvec(A, vx, vy) == 0
    dll.lis_amd_synchronize()
    t0 = time.time()
    for _ in range(reps):
        assert lib.lis_matvec(A, vx, vy) == 0
    dll.lis_amd_synchronize()
    return (time.time() - t0) / reps * 1e3
if os.environ.get("QUEEN_RUNS_AB") == "1":             # round 5: the lists as run starts (triples) against the full lists, interleaved, same process
    import hashlib
    for rep in range(3):
        for on in (1, 0):
            lib.liship_spmv_csr_set_local_runs(on)
            m = timed()
            yh = np.empty(n); lib.lis_vector_get_values(vy, 0, n, yh.ctypes.data_as(capi.P_DBL))
            print(f"runs={on}: {m:.4f} ms  y sha256 {hashlib.sha256(yh.tobytes()).hexdigest()[:16]}", flush=True)
    lib.liship_spmv_csr_set_local_runs(1)
dll.lis_amd_matrix_reordered.argtypes = [capi.PM]; dll.lis_amd_matrix_reordered.restype = C.c_longlong
reordered = dll.lis_amd_matrix_reordered(A)
if os.environ.get("QUEEN_REORDER_AB") == "1" and reordered:      # round 5: rows and columns renumbered inside the plan against the caller's numbering, interleaved
    import hashlib
    for rep in range(3):
        for on in (2, 0):                              # 2: products take the renumbered form (opt-in); 0 / 1: the caller's numbering
            lib.liship_spmv_csr_set_reorder(on)
            m = timed()
            yh = np.empty(n); lib.lis_vector_get_values(vy, 0, n, yh.ctypes.data_as(capi.P_DBL))
            print(f"reorder={on}: {m:.4f} ms  y sha256 {hashlib.sha256(yh.tobytes()).hexdigest()[:16]}", flush=True)
    lib.liship_spmv_csr_set_reorder(1)
if os.environ.get("QUEEN_PAIRS_AB") == "1":            # round 5: pairs of entries per lane (4 B position loads, 16 B LDS accesses) against single entries, interleaved
    import hashlib
    for rep in range(3):
        for on in (1, 0):
            lib.liship_spmv_csr_set_local_pairs(on)
            m = timed()
            yh = np.empty(n); lib.lis_vector_get_values(vy, 0, n, yh.ctypes.data_as(capi.P_DBL))
            print(f"pairs={on}: {m:.4f} ms  y sha256 {hashlib.sha256(yh.tobytes()).hexdigest()[:16]}", flush=True)
    lib.liship_spmv_csr_set_local_pairs(1)
if os.environ.get("QUEEN_CHAIN") == "1" and reordered:         # a caller's own iteration: y = A x, x = y * c, ... -- x is fresh from the kernel before (cache-resident), not 3 GB old
    def chain():
        va, vb = vx, vy
        for it in range(reps + 10):
            if it == 10:
                dll.lis_amd_synchronize(); t0 = time.time()
            assert lib.lis_matvec(A, va, vb) == 0
            assert lib.lis_vector_scale(0.01, vb) == 0
            va, vb = vb, va
        dll.lis_amd_synchronize()
        return (time.time() - t0) / reps * 1e3
    for rep in range(3):
        for on in (2, 0):
            lib.liship_spmv_csr_set_reorder(on)
            lisdrv.set_vector(lib, vx, xs)
            print(f"chain (product + scale) reorder={on}: {chain():.4f} ms per step", flush=True)
    lib.liship_spmv_csr_set_reorder(1)
    lisdrv.set_vector(lib, vx, xs)
ms = timed()
print(json.dumps({"n": n, "nnz": nnz, "generate_s": round(t_gen, 2), "lis_input_s": round(t_read, 2), "block_local_columns_listed": int(listed), "listed_after_reordering": int(reordered), "plan_s": round(t_plan, 2),
                  "spmv_ms": round(ms, 4), "spmv_gflops": round(2.0 * nnz / ms / 1e6, 1),
                  "frac_of_8TBs_on_contract_bytes": round((12.0 * nnz + 20.0 * n) / (ms * 1e-3) / 8e12, 4)}), flush=True)
