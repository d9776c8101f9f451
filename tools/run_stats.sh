# one rocprofv3 --kernel-trace --stats pass of bench.py; summary to gpurun_out/<tag>/stats.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 60 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | head -30 | cut -c1-220
