"""Where the host side of a plan build goes: reads the rocprofv3 --hip-trace --kernel-trace CSVs of `tools/traffic_child.py N 1 1` (tools/plan_trace.sh) and prints, for
the window between the generator kernel and the first product kernel, the HIP API calls by total time, the calls longer than 0.5 ms in order, and the kernels.
    python tools/plan_trace_summary.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main():
    d = sys.argv[1]
    api, ker = find(d, "*hip_api_trace.csv"), find(d, "*kernel_trace.csv")
    if not api or not ker:
        print("no traces under", d)
        return
    K = sorted(rows(ker), key=lambda r: int(r["Start_Timestamp"]))
    A = sorted(rows(api), key=lambda r: int(r["Start_Timestamp"]))
    gen = [k for k in K if "poisson3d" in k["Kernel_Name"]]
    prod = [k for k in K if k["Kernel_Name"].startswith("spmv_csr") or "void spmv_csr" in k["Kernel_Name"]]
    t0 = int(gen[-1]["End_Timestamp"]) if gen else int(K[0]["Start_Timestamp"])
    t1 = int(prod[0]["Start_Timestamp"]) if prod else int(K[-1]["End_Timestamp"])
    print(f"window: end of the generator kernel -> first product kernel = {(t1 - t0) / 1e6:.1f} ms")
    by = defaultdict(lambda: [0, 0])
    long_calls = []
    for r in A:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e < t0 or s > t1:
            continue
        by[r["Function"]][0] += e - s
        by[r["Function"]][1] += 1
        if e - s > 500000:
            long_calls.append((s - t0, e - s, r["Function"]))
    print("HIP API inside the window (total ms, calls):")
    for name, (ns, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {ns / 1e6:9.2f} ms {c:6d}  {name}")
    print(f"  sum of API time {sum(v[0] for v in by.values()) / 1e6:.1f} ms")
    print("calls > 0.5 ms, in order (at ms, took ms):")
    for at, took, name in long_calls:
        print(f"  {at / 1e6:8.2f} {took / 1e6:8.2f}  {name}")
    print("kernels inside the window (at ms, took ms):")
    tot = 0
    for k in K:
        s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
        if s < t0 or s >= t1:
            continue
        tot += e - s
        print(f"  {(s - t0) / 1e6:8.2f} {(e - s) / 1e6:8.3f}  {k['Kernel_Name'][:110]}")
    print(f"  kernel time {tot / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
