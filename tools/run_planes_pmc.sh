cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/planespmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for f in default planes; do
  export DOM_FORMS=$f
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $OUT/pmc_$f -o pmc -- python $GRAFT_REPO_ROOT/tools/dom_probe.py 512 1 > $OUT/pmc_$f.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT | grep -E "valuerec_dom_kernel" | cut -c1-300
