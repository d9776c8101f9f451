#!/bin/bash
# rocprofv3 kernel times of the Queen-class product: the caller's (scrambled) numbering with the plan's reordering on / off, and the natural numbering
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/queenstats; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for tag in reorder noreorder natural; do
  case $tag in reorder) E="";; noreorder) E="LIS_AMD_NO_REORDER=1";; natural) E="QUEEN_BAND=1";; esac
  env $E timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o q -- python $GRAFT_REPO_ROOT/tools/queen_probe.py 50 > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print("   ", r["Name"][:90], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
