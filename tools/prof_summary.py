"""Summarise a tools/prof.sh output directory: per-kernel avg duration + PMC counters per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats): " + os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            lines.append(f"{row.get('Name','')[:90]:90s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} "
                         f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} pct={row.get('Percentage')}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row.get("Kernel_Name", "")[:110], row.get("Counter_Name"))
                agg[key][0] += float(row.get("Counter_Value", 0))
                agg[key][1] += 1
        lines.append("== PMC: " + os.path.relpath(f, out))
        for (k, c), (v, cnt) in sorted(agg.items()):
            lines.append(f"{k:110s} {c:28s} per-launch={v/cnt:.6g} launches={cnt}")
txt = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
print(txt)
