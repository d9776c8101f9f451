cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_device_loops_gpu.py tests/test_lisapi_gpu.py -x -q > gpurun_out/r2h/pytest_k.log 2>&1
grep -E "passed|failed|rror|assert" gpurun_out/r2h/pytest_k.log | tail -8
(timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/r2h/bench.log
python - <<'PY'
import json
for f in ('bench',):
    d=json.loads(open(f'gpurun_out/r2h/{f}.log').read())
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['frac_of_stored_bytes'], d['roofline'].get('row_patterns'))
    for k,v in d['krylov'].items(): print('  ',k, v['iters_per_sec'], v['roofline']['frac'], v['roofline']['frac_of_contract_bytes'])
PY
