cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
V=0,0x40000010,0x40000810,0x40001010,0x40000040,0x40000050,0x40000000,0x40000030,0
SWEEP_VARIANTS=$V timeout 900 python tests/perf/irregular_sweep.py 80 --gmres-iters 0 2>&1 > gpurun_out/r2c/irregular_staged.log
SWEEP_VARIANTS=$V timeout 600 python tools/rowlen_sweep.py 24 32 48 80 > gpurun_out/r2c/rowlen_staged.log 2>&1
cat gpurun_out/r2c/irregular_staged.log gpurun_out/r2c/rowlen_staged.log
