#!/bin/bash
# usage: tools/prof_formats.sh <tag> <G> [fmts]   -- kernel stats + the two traffic PMC passes of tools/formats_child.py, summary into gpurun_out/<tag>/summary.txt
set -u
TAG=$1; G=$2; FMTS=${3:-csr,ell,dia,bsr}; EXTRA=${4:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
python $GRAFT_REPO_ROOT/tools/formats_child.py $G 20 $FMTS $EXTRA > $OUT/timing.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/formats_child.py $G 20 $FMTS $EXTRA > $OUT/trace.log 2>&1
for CTRS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $CTRS | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $CTRS -d $OUT/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/formats_child.py $G 12 $FMTS $EXTRA > $OUT/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -E "spmv|== " | cut -c1-260
cat $OUT/timing.txt
