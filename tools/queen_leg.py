"""bench.py's config4_stand_in leg alone (the Queen-class matrix through lis_input, lis_matvec, lis_solve): python tools/queen_leg.py"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lis_amd
import bench
lib = lis_amd.load(); assert lib.initialize([]) == 0
lib.dll.lis_amd_set_residency(1)
out = bench.queen_class_leg(lib, np, C)
print(json.dumps({k: out[k] for k in ("spmv_ms", "spmv_ms_callers_numbering", "solves") if k in out}))
