"""the 2 x 2 (b x b) blocking of the 7-point stencil through the Lis API: what the HBM copy runs on, and how fast: python tools/bsr_rowform_probe.py [N] [b]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd  # noqa: E402
import lisdrv   # noqa: E402
import orc      # noqa: E402
from lis_amd import _capi as capi, check  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
dll = lib.dll
dll.lis_amd_stream.restype = C.c_void_p
for f in ("lis_amd_matrix_value_records", "lis_amd_matrix_device_type", "lis_amd_matrix_wide_dominant", "lis_amd_matrix_row_patterns", "lis_amd_matrix_block_rows"):
    getattr(dll, f).argtypes = [capi.PM]
ptr, idx, val = orc.poisson3d(N, N, N, sort_cols=True)
n, nnz = len(ptr) - 1, len(idx)
A = lisdrv.make_csr(lib, ptr, idx, val)
x = np.modf(np.arange(n) * 0.6180339887498949)[0] - 0.5
want = None
for union, blocks in ((1, 2), (1, 1), (1, 0), (0, 0), (-1, -1)):          # (2: the block-row kernel at any size; -1: the native blocks)
    dll.lis_amd_set_row_form(0 if union < 0 else 1)
    lib.liship_spmv_csr_set_wide_union(max(union, 0))
    lib.liship_spmv_csr_set_block_rows(max(blocks, 0))
    B = lisdrv.convert(lib, lisdrv.make_csr(lib, ptr, idx, val), "bsr", b, b)
    vx, vy = lisdrv.new_vector(lib, B), lisdrv.new_vector(lib, B)
    lisdrv.set_vector(lib, vx, x)
    for _ in range(5):
        assert lib.lis_matvec(B, vx, vy) == 0
    timer = C.c_void_p()
    check(lib.liship_timer_create(C.byref(timer)))
    stream = dll.lis_amd_stream()
    check(lib.liship_timer_start(timer, stream))
    for _ in range(30):
        assert lib.lis_matvec(B, vx, vy) == 0
    check(lib.liship_timer_stop(timer, stream))
    ms = C.c_float()
    check(lib.liship_timer_elapsed_ms(timer, C.byref(ms)))
    y = lisdrv.get_vector(lib, vy, n)
    want = y if want is None else want
    print(f"N={N} {b}x{b} block-row kernel {blocks} union {union}: device type {dll.lis_amd_matrix_device_type(B)} patterns {dll.lis_amd_matrix_row_patterns(B)} value records {dll.lis_amd_matrix_value_records(B)} "
          f"wide dominant {dll.lis_amd_matrix_wide_dominant(B)} block rows {dll.lis_amd_matrix_block_rows(B)}: {ms.value / 30:.4f} ms  {2e-6 * nnz / (ms.value / 30):.0f} GFLOP/s  same bits {np.array_equal(y.view(np.uint64), want.view(np.uint64))}", flush=True)
lib.liship_spmv_csr_set_wide_union(1)
lib.liship_spmv_csr_set_block_rows(1)
dll.lis_amd_set_row_form(1)
