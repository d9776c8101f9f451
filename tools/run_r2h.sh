cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2h/pytest_all.log 2>&1
grep -E "passed|failed|rror|^FAILED" gpurun_out/r2h/pytest_all.log | tail -8
timeout 900 python tests/perf/format_sweep.py 256 --solve 2>&1 | grep -v "^$\|linear solver\|precon\|convergence\|matrix storage\|initial vector\|precision" > gpurun_out/r2h/format_sweep_256.log
cat gpurun_out/r2h/format_sweep_256.log
timeout 600 python tools/plan_time.py > gpurun_out/r2h/plan_time.log 2>&1; tail -5 gpurun_out/r2h/plan_time.log
