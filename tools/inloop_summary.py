"""per-launch counters of the kernels of a CG + Jacobi iteration, in the loop and isolated (tools/run_inloop_pmc.sh)"""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def collect(prefix):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, prefix + "_[0-9]*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    return agg
def times(prefix):
    t = {}
    for f in glob.glob(os.path.join(out, prefix + "_trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            t[row["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e6)
    return t
loop, iso, tl, ti = collect("loop"), collect("iso"), times("loop"), times("iso")
print("kernel times (rocprofv3 --stats, ms): in the loop / isolated")
for k in sorted(set(tl) | set(ti)):
    if "spmv" in k or "cg_" in k or "reduce_level1" in k:
        print(f"  {k:60s} loop {tl.get(k, (0, 0))[0]:5d} x {tl.get(k, (0, 0))[1]:.4f}    isolated {ti.get(k, (0, 0))[0]:5d} x {ti.get(k, (0, 0))[1]:.4f}")
print("counters per launch: in the loop / isolated")
for (k, c) in sorted(set(loop) | set(iso)):
    if "spmv" in k or "cg_" in k or "reduce_level1" in k:
        a, b = loop.get((k, c), [0, 1]), iso.get((k, c), [0, 1])
        print(f"  {k:60s} {c:28s} loop {a[0] / max(a[1], 1):12.5g} ({a[1]})   isolated {b[0] / max(b[1], 1):12.5g} ({b[1]})")
