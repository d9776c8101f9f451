"""Where the heavy-tailed product spends its time: the matrix of tests/perf/irregular_sweep.py whole, with its rows cut at CAP entries, and the long rows alone:
python tools/zipf_probe.py [CAP]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "perf")]
import lis_amd  # noqa: E402
import lisdrv   # noqa: E402
from irregular_sweep import zipf, time_spmv  # noqa: E402

CAP = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = lis_amd.load()
assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
lib.dll.lis_amd_stream.restype = C.c_void_p
ptr, idx, val, n = zipf(2_000_000)
lens = np.diff(ptr)
print("rows over", CAP, ":", int((lens > CAP).sum()), "with", int(lens[lens > CAP].sum()), "entries; at the cap of 200000:", int((lens == 200000).sum()), flush=True)
x = np.cos(np.arange(n) * 0.01) + 1.25


def run(name, p, i, v):
    A = lisdrv.make_csr(lib, p.astype(np.int32), i, v)
    vx, vy = lisdrv.new_vector(lib, A, x), lisdrv.new_vector(lib, A)
    ms = time_spmv(lib, lib.dll, A, vx, vy)
    print(f"{name:40s} nnz {len(i):9d}  {ms:.4f} ms", flush=True)
    lib.lis_matrix_destroy(A)


run("whole", ptr, idx, val)
keep = np.minimum(lens, CAP)
sel = np.concatenate([np.arange(ptr[r], ptr[r] + keep[r]) for r in np.flatnonzero(lens > CAP)]) if (lens > CAP).any() else np.zeros(0, np.int64)
mask = np.ones(len(idx), bool)
for r in np.flatnonzero(lens > CAP):
    mask[ptr[r] + CAP:ptr[r + 1]] = False
p2 = np.zeros(n + 1, np.int64); np.cumsum(keep, out=p2[1:])
run(f"rows cut at {CAP}", p2, idx[mask], val[mask])
long_only = np.where(lens > CAP, lens, 0)
p3 = np.zeros(n + 1, np.int64); np.cumsum(long_only, out=p3[1:])
m3 = np.zeros(len(idx), bool)
for r in np.flatnonzero(lens > CAP):
    m3[ptr[r]:ptr[r + 1]] = True
run(f"only the rows over {CAP}", p3, idx[m3], val[m3])
one = np.where(np.arange(n) == int(np.argmax(lens)), lens, 0)
p4 = np.zeros(n + 1, np.int64); np.cumsum(one, out=p4[1:])
r = int(np.argmax(lens))
run("only the longest row", p4, idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]])
