"""Golden vectors of the reference for the split form A = L + D + U.

Dev container only: oracle/_ref (Lis 2.1.11 from /root/reference/src, 1 OpenMP thread).  For two matrices with a full diagonal,
in every storage format: lis_matrix_convert -> lis_matrix_split (src/matrix/lis_matrix_ops.c:860; not in the public header, but
an exported symbol of the library) -> the arrays of A->L, A->U, A->D, and y = lis_matvec(A, x) on the split matrix (the
is_splited branches of src/matvec/lis_matvec_<fmt>.c), next to y of the unsplit matrix.
    python tests/golden/make_golden_split.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import lisdrv  # noqa: E402
import orc     # noqa: E402
from make_golden_scale import test_matrix  # noqa: E402

CASES = [("csr", 0), ("csc", 0), ("ell", 0), ("dia", 0), ("jad", 0), ("bsr", 1), ("bsr", 2), ("bsr", 3), ("bsr", 4), ("bsr", 5)]


def matrices():
    yield "p3d_6x5x4", orc.poisson3d(6, 5, 4)
    yield "nonsym_61", test_matrix(61, 5)
    ptr, idx, val = test_matrix(40, 11)
    val = val.copy()
    val[::7] = -val[::7]                       # sign changes and a few exact zeros: signed-zero behaviour of the chains
    val[3::11] = 0.0
    yield "zeros_40", (ptr, idx, val)


def main():
    orc.build()
    ref = lisdrv.open_lib(orc.REF_SO, threads=1)
    out = {}
    for name, (ptr, idx, val) in matrices():
        n = len(ptr) - 1
        x = np.sin(np.arange(n) * 0.7) + 0.25
        x[::5] = -0.0                          # -0.0 * D stays -0.0 only when the first product initialises the sum
        out[f"{name}/ptr"], out[f"{name}/idx"], out[f"{name}/val"], out[f"{name}/x"] = ptr, idx, val, x
        for fmt, bs in CASES:
            A = lisdrv.make_csr(ref, ptr, idx, val)
            B = A if fmt == "csr" else lisdrv.convert(ref, A, fmt, bs, bs)
            key = f"{name}/{fmt}{bs if bs else ''}"
            out[key + "/y_unsplit"] = lisdrv.matvec(ref, B, x)
            assert ref.lis_matrix_split(B) == 0
            parts = lisdrv.split_arrays(B)
            for tag in ("L", "U"):
                for k, v in parts[tag].items():
                    out[f"{key}/{tag}/{k}"] = np.asarray(v)
            out[key + "/D"] = parts["D"]
            out[key + "/y_split"] = lisdrv.matvec(ref, B, x)
            print(key, "split y differs from unsplit y in", int((out[key + "/y_split"] != out[key + "/y_unsplit"]).sum()), "of", n, "rows")
    np.savez_compressed(os.path.join(HERE, "split_golden.npz"), **out)


if __name__ == "__main__":
    main()
