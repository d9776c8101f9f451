# rocprofv3 --kernel-trace --stats of the format / BSR / irregular sweeps (per-kernel durations for the kernels bench.py does not launch)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/sweepstats; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
LIS_AMD_NO_VALUE_RECORDS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/formats -o t -- python $GRAFT_REPO_ROOT/tests/perf/format_sweep.py 256 > $OUT/formats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bsr -o t -- python $GRAFT_REPO_ROOT/tests/perf/bsr_sweep.py 256 > $OUT/bsr.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bsrfem -o t -- python $GRAFT_REPO_ROOT/tests/perf/bsr_sweep.py --fem 100 3 > $OUT/bsrfem.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/irregular -o t -- python $GRAFT_REPO_ROOT/tests/perf/irregular_sweep.py > $OUT/irregular.log 2>&1
cd $GRAFT_REPO_ROOT
for d in formats bsr bsrfem irregular; do
  echo "## $d: $(grep -E '^(csr|ell|dia|jad|bsr|csc|fem3|zipf)' $OUT/$d.log | cut -c1-110 | tr '\n' '|')"
  f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    if int(r["Calls"]) >= 20:
        print(f"  {r['Name'][:120]:120s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} min_us={float(r['MinNs'])/1e3:10.2f}")
PY
done
