#!/bin/bash
# usage: tools/prof.sh <outdir-under-gpurun_out> <command...>   -- kernel-trace stats + separate PMC passes
# (PMC passes are separate runs with --kernel-trace only, as the pool requires)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$tag -o pmc -- "$@" > $OUT/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT
