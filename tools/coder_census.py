"""which plan coders the random structured matrices of tests/test_kernels_gpu.py reach: python tools/coder_census.py [seeds]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lis_amd  # noqa: E402
from lis_amd import DeviceArray as DA, check  # noqa: E402

src = open(os.path.join(ROOT, "tests", "test_kernels_gpu.py")).read()
ns = {"np": np}
exec(src[src.index("def _structured_random(seed):"):src.index('@pytest.mark.parametrize("seed", range(int(os.environ')], ns)
lib = lis_amd.load()
census = collections.Counter()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    ptr, idx, val, ncols = ns["_structured_random"](seed)
    n = len(ptr) - 1
    dptr = DA.from_host(ptr, np.int32)
    didx = DA.from_host(idx if len(idx) else np.zeros(1, np.int32), np.int32)
    dval = DA.from_host(val if len(val) else np.zeros(1), np.float64)
    plan = C.c_void_p()
    check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, None))
    check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, None))
    check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, None))
    before = lib.liship_csr_plan_row_patterns(plan)
    check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, None))
    after = lib.liship_csr_plan_row_patterns(plan)
    census[(lib.liship_csr_plan_coded(plan) > 0, before > 0, lib.liship_csr_plan_pattern_records(plan), lib.liship_csr_plan_value_records(plan),
            "refined" if after != before else "")] += 1
    check(lib.liship_csr_plan_destroy(plan))
for k, v in sorted(census.items(), key=lambda kv: -kv[1]):
    print(v, "coded=%s patterns=%s records=%s value_records=%s %s" % k)
