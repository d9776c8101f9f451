# rocprofv3 counters of the BSR kernels (tests/perf/bsr_sweep.py 200: 2x2, 3x3, 4x4 blocks of the 200^3 stencil), separate passes per group
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/bsrpmc; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/tests/perf/bsr_sweep.py ${BSR_ARGS:-200} > $OUT/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -E "bsr" | cut -c1-70,110-240
