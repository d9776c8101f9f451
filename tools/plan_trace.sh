#!/bin/bash
# usage: tools/plan_trace.sh <outdir-under-gpurun_out> [N]   -- the host side of a plan build: HIP API + kernel timeline of one matrix + plan + product (no counters)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=${2:-512}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/traffic_child.py $N 1 1 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/plan_trace_summary.py $OUT/trace
