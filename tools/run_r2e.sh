cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2e/pytest_all.log 2>&1
grep -E "passed|failed|rror|^FAILED|assert" gpurun_out/r2e/pytest_all.log | tail -20
