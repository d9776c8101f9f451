set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r2a/pytest.log 2>&1

timeout 1500 tools/prof.sh r2a/prof python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r2a/prof.log 2>&1
tail -3 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench.log
