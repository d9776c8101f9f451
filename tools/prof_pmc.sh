#!/bin/bash
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> "<counters...>" <command...>   -- one PMC pass (kernel-trace only)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CTRS="$1"; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
tag=$(echo $CTRS | tr ' ' '_')
rocprofv3 --kernel-trace --output-format csv --pmc $CTRS -d $OUT/pmc_$tag -o pmc -- "$@" > $OUT/pmc_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -i "spmv" 
