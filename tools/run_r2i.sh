cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
for v in "" "LIS_AMD_NO_ROW_PATTERNS=1" "LIS_AMD_NO_INDEX_CODES=1"; do echo "## $v"; env $v timeout 300 python tools/plan_time.py 2>&1 | tail -2; done > gpurun_out/r2i/plan_time.log
cat gpurun_out/r2i/plan_time.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r2i/bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2i/bench.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['frac_of_stored_bytes'], d['cpu_baseline']['value'])
for k,v in d['krylov'].items(): print('  ',k, v['iters_per_sec'], v['roofline']['frac'], v['roofline']['frac_of_contract_bytes'])
PY
PROF_PASS_TIMEOUT=300 timeout 2400 tools/prof.sh r2i/prof python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r2i/prof.log 2>&1
grep "pattern_kernel<256, 2048, 7, 0" gpurun_out/r2i/prof/summary.txt | cut -c1-60,150-260
