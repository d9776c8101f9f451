cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests/test_configs_gpu.py -q > gpurun_out/r2e/pytest_configs.log 2>&1
grep -E "passed|failed|rror|assert|Error" gpurun_out/r2e/pytest_configs.log | tail -30
