cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $OUT
timeout 900 python tests/perf/format_sweep.py 256 --solve 2>&1 | grep -v "^$\|linear solver\|precon\|convergence\|matrix storage\|initial vector\|precision" > $OUT/format_sweep_256.log
timeout 600 python tests/perf/bsr_sweep.py 256 > $OUT/bsr_sweep.log 2>&1
timeout 900 python tests/perf/irregular_sweep.py 80 --gmres-iters 150 2>&1 | grep -v "^$\|linear solver\|precon\|convergence\|matrix storage\|initial vector\|precision" > $OUT/irregular.log
export TMPDIR=/tmp IRREG_ONLY=fem3
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tests/perf/irregular_sweep.py 80 --gmres-iters 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT" \
         "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$i -o pmc -- $CMD > $OUT/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -i "spmv_csr\|==" | cut -c1-60,100-260 > $OUT/fem3_pmc.txt
cat $OUT/format_sweep_256.log $OUT/bsr_sweep.log $OUT/irregular.log $OUT/fem3_pmc.txt
