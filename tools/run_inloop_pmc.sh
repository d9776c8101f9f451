#!/bin/bash
# Why the fused dominant-pattern product takes 0.60 ms inside lis_solve and 0.50 ms launched back to back:
# L2 / fabric counters per launch of spmv_csr_valuerec_dom_kernel<256, 1> in the CG + Jacobi loop (tools/solve512_short.py) and in the
# isolated probe (tools/dom_probe.py), separate rocprofv3 passes per counter group (kernel-trace only).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/inloop; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export DOM_FORMS="default"
i=0
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/loop_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/solve512_short.py > $OUT/loop_$i.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/iso_$i -o pmc -- python $GRAFT_REPO_ROOT/tools/dom_probe.py 512 1 > $OUT/iso_$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop_trace -o t -- python $GRAFT_REPO_ROOT/tools/solve512_short.py > $OUT/loop_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso_trace -o t -- python $GRAFT_REPO_ROOT/tools/dom_probe.py 512 1 > $OUT/iso_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/inloop_summary.py $OUT
