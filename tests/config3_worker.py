"""BASELINE config 3's exact partition -- the ONE 512^3 Poisson grid, row blocks over 8 ranks, 64 planes each -- with all the
ranks on one GPU over the host-callback communicator (RCCL refuses ranks that share a device; the partition, the halo tables, the overlapped
product and the rank-order folds are the ones an 8-GPU RCCL job runs).  Launched by tests/test_config3_gpu.py, once with WORLD_SIZE=8 and once
with WORLD_SIZE=1 (the single-rank job the slices are compared with).  Every rank prints one JSON line:

  tables     neighbours, export / import counts and bytes per neighbour (2 x 2 MiB out, 2 x 2 MiB in for interior ranks)
  y_sha256   sha256 of this rank's slice of y = A x, x_i = frac(i * golden ratio) - 0.5  (world 1: of each of the 8 slices)
  slabs      this rank's first / middle / last 2^18 rows of y against the ORACLE's product of the oracle-generated rows (bits)
  bicgstab   -i bicgstab -p none to 1e-12 on b = A*1: count, status, residual (tree reductions)
  ref_hist   the first 24 BiCGSTAB iterations (CONFIG3_REF_MAXITER) in the reference-order mode: T = 1 per rank in the 8-rank job, T = 8 in the single-rank job --
             chunks by LIS_GET_ISIE are the ranks' row blocks and the fold is in rank order, so the two histories must agree IN EVERY BIT
"""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import lis_amd  # noqa: E402
import lisdrv  # noqa: E402
import orc  # noqa: E402
from lis_amd import _capi as capi  # noqa: E402
from lis_amd._hostcomm import Callbacks, make_callbacks  # noqa: E402

N = int(os.environ.get("CONFIG3_N", "512"))
PARTS = 8
GOLD = 0.6180339887498949


def xvals(lo, hi):
    return np.modf(np.arange(lo, hi, dtype=np.float64) * GOLD)[0] - 0.5


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lib = lis_amd.load()
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cb = make_callbacks(world)
        lib.dll.lis_amd_comm_init_callbacks.argtypes = [C.POINTER(Callbacks), C.c_int, C.c_int]
        assert lib.dll.lis_amd_comm_init_callbacks(C.byref(cb), rank, world) == 0
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(1)
    if os.environ.get("CONFIG3_NO_OVERLAP") == "1":     # A/B: the whole local product in one launch behind the halo exchange
        lib.dll.lis_amd_set_overlap(0)
    if os.environ.get("CONFIG3_DOM_MARCH"):            # A/B: the form of the dominant-pattern product (liship_spmv_csr_set_dom_march)
        lib.liship_spmv_csr_set_dom_march(int(os.environ["CONFIG3_DOM_MARCH"]))
    gn, mn = N ** 3, N * N
    A = capi.PM()
    assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0
    assert lib.lis_matrix_set_size(A, 0, gn) == 0
    lib.dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
    assert lib.dll.lis_amd_matrix_poisson3d(A, N, N, N, 0) == 0
    a = A.contents
    is_, ie, nl = a.is_, a.ie, a.n
    out = {"rank": rank, "world": world, "is": is_, "ie": ie, "n": nl, "np": a.np}
    if world > 1:
        assert nl == gn // world and is_ == rank * nl, (is_, ie)              # 64 planes per rank at 512^3 / 8
        t = a.commtable.contents
        nb = t.neibpetot
        out["tables"] = {"neighbours": list(t.neibpe[:nb]),
                         "export_counts": [t.export_ptr[i + 1] - t.export_ptr[i] for i in range(nb)],
                         "import_counts": [t.import_ptr[i + 1] - t.import_ptr[i] for i in range(nb)],
                         "export_first_rows": [int(t.export_index[t.export_ptr[i]]) for i in range(nb)],
                         "send_bytes": 8 * int(t.export_ptr[nb]), "recv_bytes": 8 * int(t.import_ptr[nb])}
    # ---- y = A x, slices and slabs
    vx, vy, vb, vs = (lisdrv.new_vector(lib, A, None) for _ in range(4))
    xl = xvals(is_, ie)
    assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, is_, nl, xl.ctypes.data_as(capi.P_DBL), vx) == 0
    assert lib.lis_matvec(A, vx, vy) == 0
    y = np.empty(nl)
    assert lib.lis_vector_get_values(vy, is_, nl, y.ctypes.data_as(capi.P_DBL)) == 0
    part = gn // PARTS
    if world > 1:
        out["y_sha256"] = [hashlib.sha256(y.tobytes()).hexdigest()]
    else:
        out["y_sha256"] = [hashlib.sha256(y[p * part:(p + 1) * part].tobytes()).hexdigest() for p in range(PARTS)]
    slab = min(1 << 18, nl)
    checked = []
    for r0 in sorted({is_, is_ + (nl - slab) // 2 + 77 if nl > slab + 77 else is_, ie - slab}):
        r1 = r0 + slab
        ptr_s, idx_s, val_s = orc.poisson3d(N, N, N, is_=r0, ie=r1)
        lo, hi = max(0, r0 - mn), min(gn, r1 + mn)
        yref = orc.spmv_csr(ptr_s, idx_s - lo, val_s, xvals(lo, hi))
        assert np.array_equal(y[r0 - is_:r1 - is_].view(np.uint64), yref.view(np.uint64)), ("slab", rank, r0)
        checked.append(int(r0))
    out["slabs"] = checked
    # ---- BiCGSTAB on b = A*1 (closed form), tree reductions, to 1e-12
    lib.dll.lis_amd_vector_poisson3d_rhs.argtypes = [capi.PV, C.c_int, C.c_int, C.c_int]
    assert lib.dll.lis_amd_vector_poisson3d_rhs(vb, N, N, N) == 0

    def solve(opts):
        S = capi.PS()
        assert lib.lis_solver_create(C.byref(S)) == 0
        assert lib.lis_solver_set_option(opts.encode(), S) == 0
        assert lib.lis_solve(A, vb, vs, S) == 0
        it, res, st = C.c_int(), C.c_double(), C.c_int()
        lib.lis_solver_get_iter(S, C.byref(it)); lib.lis_solver_get_residualnorm(S, C.byref(res)); lib.lis_solver_get_status(S, C.byref(st))
        maxiter = S.contents.options[2]
        hist = [float(S.contents.rhistory[i]).hex() for i in range(min(it.value, maxiter) + 1)]
        itime = S.contents.itime
        lib.lis_solver_destroy(S)
        return it.value, st.value, res.value, hist, itime

    it, st, res, _, itime = solve("-i bicgstab -p none -tol 1e-12 -maxiter 3000")
    xs = np.empty(nl)
    assert lib.lis_vector_get_values(vs, is_, nl, xs.ctypes.data_as(capi.P_DBL)) == 0
    out["bicgstab"] = {"iter": it, "status": st, "resid": res, "itime": round(itime, 3), "max_err": float(np.abs(xs - 1.0).max())}
    # ---- the same recurrence in the reference's summation order: 8 ranks x 1 chunk == 1 rank x 8 chunks
    assert lib.dll.lis_amd_set_reference_reductions(1 if world > 1 else PARTS) == 0
    it, st, res, hist, _ = solve("-i bicgstab -p none -tol 1e-12 -maxiter %d -print mem" % int(os.environ.get("CONFIG3_REF_MAXITER", "24")))
    lib.dll.lis_amd_set_reference_reductions(0)
    out["ref_hist"] = {"iter": it, "status": st, "rhistory": hist}
    if world > 1:
        dist.barrier()
    print("CONFIG3 " + json.dumps(out), flush=True)
    for v in (vx, vy, vb, vs):
        lib.lis_vector_destroy(v)
    lib.lis_matrix_destroy(A)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
