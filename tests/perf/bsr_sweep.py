"""BSR products of the 256^3 stencil with 2x2, 3x3, 4x4 blocks (lis_matvec through the C API), against the bytes the
format stores.   python tests/perf/bsr_sweep.py [N]        python tests/perf/bsr_sweep.py --fem G [dofs]: the dofs-per-node
FEM pattern of a G^3 grid (27 dofs x dofs blocks per block row: the long-block-row case) in dofs x dofs blocks"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
import lisdrv   # noqa: E402
import orc      # noqa: E402
from lis_amd import _capi as capi  # noqa: E402


def main():
    fem = len(sys.argv) > 2 and sys.argv[1] == "--fem"
    N = int(sys.argv[1]) if len(sys.argv) > 1 and not fem else 256
    lib = lis_amd.load()
    assert lib.initialize([]) == 0
    lib.dll.lis_amd_set_residency(1)
    dofs = int(sys.argv[3]) if fem and len(sys.argv) > 3 else 3
    ptr, idx, val = orc.fem3(int(sys.argv[2]), dofs)[:3] if fem else orc.poisson3d(N, N, N)
    n, nnz = len(ptr) - 1, len(idx)
    x = np.ones(n)
    for bs in ((dofs,) if fem else (2, 3, 4)):
        A = lisdrv.make_csr(lib, ptr, idx, val)
        B = lisdrv.convert(lib, A, "bsr", bs, bs)
        bnnz, nr = B.contents.bnnz, B.contents.nr
        vx, vy = lisdrv.new_vector(lib, B, x), lisdrv.new_vector(lib, B)
        for team in (1, 0):                     # 0: long block rows through the two-phase tile kernels (A/B)
            lib.liship_spmv_bsr_set_team(team)
            for _ in range(20):
                assert lib.lis_matvec(B, vx, vy) == 0
            lib.dll.lis_amd_synchronize()
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                assert lib.lis_matvec(B, vx, vy) == 0
            lib.dll.lis_amd_synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            alg = bnnz * (8 * bs * bs + 4) + 4 * nr + 16 * n
            nrm = C.c_double()
            lib.lis_vector_nrm2(vy, C.byref(nrm))
            print(f"bsr {bs}x{bs} team {team}: {bnnz / nr:.1f} blocks/row  {ms:.4f} ms  {2 * nnz / ms / 1e6:.1f} GFLOP/s  {alg / ms / 1e6:.0f} GB/s alg "
                  f"({alg / ms / 1e6 / 80:.1f}% of 8 TB/s)  ||A*1||={nrm.value:.6e}", flush=True)
        lib.liship_spmv_bsr_set_team(1)
        lib.lis_matrix_destroy(B)


if __name__ == "__main__":
    main()
