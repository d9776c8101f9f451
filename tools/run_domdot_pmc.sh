#!/bin/bash
# rocprofv3 counters of the fused-dot forms of the value-record product at 512^3 (tools/dom_probe.py), separate passes per group (kernel-trace only)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/domdotpmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export DOM_FORMS="default"
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/dom_probe.py 512 1 > $OUT/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -E "valuerec_dom" | cut -c1-75,112-240
