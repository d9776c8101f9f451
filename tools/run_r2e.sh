cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests/test_device_loops_gpu.py tests/test_lisapi_gpu.py tests/test_rccl_world1_gpu.py tests/test_more_solvers_gpu.py -x -q > gpurun_out/r2e/pytest_full.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r2e/pytest_full.log | tail -5
