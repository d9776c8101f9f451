import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np
import lis_amd, lisdrv, orc
sys.path.insert(0, "tests")
from test_lisapi_gpu import _nonsym_dominant
lib = lis_amd.load(); lib.initialize([])
ptr, idx, val = _nonsym_dominant(500, 31)
n = 500
b = orc.spmv_csr(ptr, idx, val, np.ones(n))
A = lisdrv.make_csr(lib, ptr, idx, val)
out = lisdrv.solve(lib, A, b, "-i bicg -p none -tol 1e-12 -maxiter 40 -print mem")
x, it, rc, resid, rh = orc.bicg(ptr, idx, val, b, tol=1e-12, maxiter=40)
print("oracle", it, rc, rh[:10])
print("amd   ", out["iter"], out["status"], out["rhistory"][:10])
