#!/bin/bash
# counters of the block-local kernel on the Queen-class mesh: the plan's Cuthill-McKee numbering (products opted in) against the mesh's natural numbering (QUEEN_BAND=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
for tag in cm natural; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/queennat_$tag; rm -rf $OUT; mkdir -p $OUT
  E="LIS_AMD_REORDER_PRODUCTS=1"; [ $tag = natural ] && E="QUEEN_BAND=1"
  echo "== $tag"
  env $E python $GRAFT_REPO_ROOT/tools/queen_probe.py 50 2>&1 | tail -1
  i=0
  for c in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    env $E timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/queen_probe.py 20 > $OUT/pmc_$i.log 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT | grep -E "spmv_csr_local|reorder_gather" | cut -c1-70,110-400
done
