cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2f/pytest_all.log 2>&1
grep -E "passed|failed|rror|^FAILED" gpurun_out/r2f/pytest_all.log | tail -8
timeout 900 python tests/perf/format_sweep.py 256 --solve > gpurun_out/r2f/format_sweep_256.log 2>&1
grep -v "^$\|linear solver\|precon\|convergence\|matrix storage\|initial vector\|precision" gpurun_out/r2f/format_sweep_256.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r2f/bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2f/bench.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_stored_bytes'], d['cpu_baseline'])
for k,v in d['krylov'].items(): print(k, v['iters_per_sec'], v['roofline']['frac'], v['roofline']['frac_of_contract_bytes'])
PY
PROF_PASS_TIMEOUT=300 timeout 2400 tools/prof.sh r2f/prof python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r2f/prof.log 2>&1
tail -2 gpurun_out/r2f/prof.log | cut -c1-200
