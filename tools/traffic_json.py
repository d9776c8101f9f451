"""profiles/rNN_spmv512_traffic.json from a tools/prof.sh output directory.

    python tools/traffic_json.py gpurun_out/<dir> <kernel-name-substring> <out.json> [--n N --nnz NNZ --coded 0|1]

Bytes at the L2 <-> fabric boundary, per launch of the named kernel, from the request counters gfx950 keeps per size
(TCC_EA0_RDREQ_{32B,64B,128B}_sum, TCC_EA0_WRREQ_{,64B}_sum; when the DRAM-destined 32 B-granular counters are
there they are reported next to them).  These requests include the ones the 256 MB Infinity Cache answers, so the
rate can exceed what HBM itself delivers; the streams (value / code / ptr arrays, read once with nt loads) are
compulsory, the rest of the reads is x, and x_refetch_factor = x bytes / (8 B per row).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("kernel"); ap.add_argument("out")
ap.add_argument("--n", type=int, default=512 ** 3)
ap.add_argument("--nnz", type=int, default=7 * 512 ** 3 - 6 * 512 * 512)
ap.add_argument("--coded", type=int, default=1, help="0: the contract form (4 B indices: 12 B per non-zero + the row pointers)")
ap.add_argument("--patterns", type=int, default=0, help="row patterns: 8 B per non-zero + 1 B per row of matrix streams (row starts by scan)")
ap.add_argument("--values", type=int, default=0, help="value records: one pattern byte per row is the only matrix stream")
ap.add_argument("--box", type=int, default=0, help="with --values: the z-marching kernel's box form, which reads no pattern byte either (x and y alone)")
ap.add_argument("--command", default="python bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline")
a = ap.parse_args()

ctr = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(a.dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if a.kernel in row.get("Kernel_Name", ""):
                c = ctr[row["Counter_Name"]]
                c[0] += float(row["Counter_Value"]); c[1] += 1
per = {k: v[0] / v[1] for k, v in ctr.items() if v[1]}
avg_ns, launches = None, None
for f in glob.glob(os.path.join(a.dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if a.kernel in row.get("Name", ""):
                avg_ns, launches = float(row["AverageNs"]), int(row["Calls"])
                break
g = per.get
rd = None
if g("TCC_EA0_RDREQ_sum") is not None and g("TCC_EA0_RDREQ_128B_sum") is not None:
    r32, r64, r128 = g("TCC_EA0_RDREQ_32B_sum", 0.0), g("TCC_EA0_RDREQ_64B_sum", 0.0), g("TCC_EA0_RDREQ_128B_sum", 0.0)
    other = g("TCC_EA0_RDREQ_sum") - r32 - r64 - r128          # should be ~0 when the three sizes partition the requests
    rd = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, other)
wr = None
if g("TCC_EA0_WRREQ_sum") is not None:
    w64 = g("TCC_EA0_WRREQ_64B_sum", 0.0)
    wr = 64 * w64 + 32 * (g("TCC_EA0_WRREQ_sum") - w64)
n, nnz = a.n, a.nnz
stream_rd = (9 if a.coded else 12) * nnz + 4 * (n + 1)           # value + code/index + ptr, each read once
if a.patterns:
    stream_rd = 8 * nnz + n                                       # value + one pattern byte per row
if a.patterns and a.values:
    stream_rd = 0 if a.box else n                                 # one pattern byte per row (the box form: none)
out = {
    "source": f"rocprofv3 --kernel-trace --pmc <group> -- {a.command} (separate passes per counter group, tools/prof.sh), summarised by tools/traffic_json.py",
    "kernel": a.kernel, "launches": launches, "avg_kernel_ns": avg_ns,
    "level": "L2 <-> fabric requests (Infinity Cache hits included: an upper bound of the HBM bytes)",
    "counters_per_launch": {k: per[k] for k in sorted(per)},
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
    "fabric_bytes_per_launch": (rd + wr) if (rd is not None and wr is not None) else None,
    "FETCH_SIZE_x2_bytes": 2 * 1024 * g("FETCH_SIZE") if g("FETCH_SIZE") is not None else None,
    "WRITE_SIZE_bytes": 1024 * g("WRITE_SIZE") if g("WRITE_SIZE") is not None else None,
    "dram_read_bytes_32B_granular": 32 * g("TCC_EA0_RDREQ_DRAM_32B_sum") if g("TCC_EA0_RDREQ_DRAM_32B_sum") is not None else None,
    "dram_write_bytes_32B_granular": 32 * g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") if g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") is not None else None,
    "algorithmic_bytes_per_launch": 12 * nnz + 20 * n + 4,
    "stored_bytes_per_launch": ((16 if a.box else 17) * n + 4) if (a.patterns and a.values) else (8 * nnz + 17 * n + 4) if a.patterns else ((9 if a.coded else 12) * nnz + 20 * n + 4),
    "stream_read_bytes_per_launch": stream_rd,
}
if rd is not None:
    xb = rd - stream_rd
    out["x_bytes_per_launch"] = xb
    out["x_refetch_factor"] = round(xb / (8.0 * n), 3)
    if avg_ns and wr is not None:
        out["fabric_rate_GBs"] = round((rd + wr) / avg_ns, 1)
        out["traffic_over_stored"] = round((rd + wr) / out["stored_bytes_per_launch"], 4)
        cap = 8.0e12 * avg_ns * 1e-9
        out["hbm_bytes_bounds"] = [out["stored_bytes_per_launch"], int(min(rd + wr, cap))]
        out["note"] = ("exact bytes at the L2 <-> fabric boundary (requests counted by size; 2 x FETCH_SIZE agrees within 0.3 %). "
                       "They include re-reads of x that the Infinity Cache serves, which is how the rate at this boundary can pass the "
                       "8 TB/s of HBM itself: the HBM bytes lie between the compulsory stored bytes and min(fabric bytes, 8 TB/s x kernel "
                       "time) = hbm_bytes_bounds.  the matrix streams (values, codes or row patterns, row starts) are read once (nt loads): everything else on the read side "
                       "is x, fetched x_refetch_factor times across the fabric (the 8 XCD L2s each fetch their own copy of a line: the "
                       "+-n neighbours of a row block run on other XCDs, the +-mn ones 128 blocks later on the same one).")
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))
