cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2i/pytest_all.log 2>&1
grep -E "passed|failed|rror|^FAILED" gpurun_out/r2i/pytest_all.log | tail -8
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r2i/bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2i/bench.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['frac_of_stored_bytes'], d['cpu_baseline']['value'])
for k,v in d['krylov'].items(): print('  ',k, v['iters_per_sec'], v['roofline']['frac'], v['roofline']['frac_of_contract_bytes'])
PY
PROF_PASS_TIMEOUT=300 timeout 2400 tools/prof.sh r2i/prof python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r2i/prof.log 2>&1
grep "pattern_kernel<256, 2048, 7, 0" gpurun_out/r2i/prof/summary.txt | cut -c1-60,150-260 | head -4
timeout 900 python tests/perf/format_sweep.py 256 --solve 2>&1 | grep -v "^$\|linear solver\|precon\|convergence\|matrix storage\|initial vector\|precision" > gpurun_out/r2i/format_sweep_256.log
cat gpurun_out/r2i/format_sweep_256.log
