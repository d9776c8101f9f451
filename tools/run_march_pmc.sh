#!/bin/bash
# rocprofv3 counters of the headline product at 512^3: the z-marching kernel and, beside it, the gathering dominant-pattern kernel (DOM_MARCH=0), separate passes per group
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/marchpmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for m in 1 0; do
i=0
for c in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  DOM_MARCH=$m DOM_FORMS=default timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/m${m}_pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/dom_probe.py 512 1 > $OUT/m${m}_pmc_$i.log 2>&1
done
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -E "valuerec_(march|dom)_kernel" | cut -c1-300 > gpurun_out/march_pmc_summary.txt
wc -l gpurun_out/march_pmc_summary.txt
