#!/bin/bash
# usage: tools/prof.sh <outdir-under-gpurun_out> <command...>   -- kernel-trace stats + separate PMC passes
# (PMC passes are separate runs with --kernel-trace only, as the pool requires).  Set PROF_PMC_ONLY="A B|C" to choose
# the counter groups (groups separated by '|').
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout ${PROF_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
# request sizes at the L2 <-> fabric boundary are counted separately on gfx950 (32 / 64 / 128 B), so read and write bytes
# need no calibration; FETCH_SIZE / WRITE_SIZE are kept for the guide's x2 cross-check
GROUPS_DEFAULT="FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum|TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum|TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum|TCC_HIT_sum TCC_MISS_sum"
IFS='|' read -ra GR <<< "${PROF_PMC_ONLY:-$GROUPS_DEFAULT}"
for c in "${GR[@]}"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  timeout ${PROF_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$tag -o pmc -- "$@" > $OUT/pmc_$tag.log 2>&1   # a pass that hangs must not take the box with it
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT
