"""How far the REFERENCE's own iteration counts move with its reduction order: the solves of lis_ref_golden.npz that are not CG (CG's
count is stable), run by oracle/_ref (Lis 2.1.11 compiled from /root/reference by oracle/Makefile) at 1 .. 8 OpenMP threads -- each
thread count is a different grouping of the dot products' partial sums, exactly what separates this library's tree reductions from
the 1-thread reference.  tests/test_lisapi_gpu.py demands a count inside that spread (+- 1).

    python tests/golden/make_golden_spread.py        (dev container only: needs oracle/_ref; rewrites iteration_spread.json)
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

WORKER = r'''
import json, os, sys
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lisdrv, orc
threads = int(sys.argv[1])
G = np.load(os.path.join(%(here)r, "lis_ref_golden.npz"))
ref = lisdrv.open_lib(orc.REF_SO, threads=threads)
out = {}
for name in sorted({k.split("/")[1] for k in G.files if k.startswith("solve/")}):
    solver, precon = name.split("_")[0], name.split("_")[1]
    if solver == "cg":
        continue
    grid = tuple(int(v) for v in G["solve/%%s/grid" %% name])
    ptr, idx, val = orc.poisson3d(*grid)
    A = lisdrv.make_csr(ref, ptr, idx, val)
    opts = "-i %%s -p %%s -tol 1e-12 -maxiter 1000" %% (solver, precon)
    if solver == "gmres":
        opts += " -restart " + name.split("_r")[-1]
    res = lisdrv.solve(ref, A, G["solve/%%s/b" %% name], opts)
    out[name] = [int(res["iter"]), int(res["status"])]
print(json.dumps(out))
'''

if __name__ == "__main__":
    spread = {}
    for threads in range(1, 9):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        txt = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT, "here": HERE}, str(threads)], capture_output=True, text=True, env=env, check=True).stdout
        for name, (it, st) in json.loads(txt.strip().splitlines()[-1]).items():
            assert st == 0, (name, threads, st)
            spread.setdefault(name, []).append(it)
    out = {"_source": "Lis 2.1.11 (oracle/_ref) at 1..8 OpenMP threads: iteration counts per thread count, same inputs as lis_ref_golden.npz",
           "counts": spread}
    json.dump(out, open(os.path.join(HERE, "iteration_spread.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(spread.items()):
        print(k, v)
