"""what the plans choose for the random stencils of tests/test_kernels_gpu.py::_stencil_random (300 seeds): python tools/fuzz_stats.py"""
import sys, os, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, lis_amd
from lis_amd import DeviceArray as DA, check
import test_kernels_gpu as T
lib = lis_amd.load()
cnt = collections.Counter()
for seed in range(300):
    ptr, idx, val = T._stencil_random(seed)
    n = len(ptr) - 1
    dptr, didx, dval = DA.from_host(ptr, np.int32), DA.from_host(idx, np.int32), DA.from_host(val, np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    cnt[(("pat" if lib.liship_csr_plan_row_patterns(plan) else "nopat"), "rec8" if lib.liship_csr_plan_pattern_records(plan) else "", "team%d" % lib.liship_csr_plan_team_form(plan), "vrec%d" % lib.liship_csr_plan_value_records(plan), "wide%d" % lib.liship_csr_plan_wide_dominant(plan), "dom%d" % lib.liship_csr_plan_dominant_pattern(plan))] += 1
    check(lib.liship_csr_plan_destroy(plan))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(v, k)
