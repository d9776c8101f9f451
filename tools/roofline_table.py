"""Per-kernel roofline table from a rocprofv3 --kernel-trace --stats CSV of bench.py at 512^3 (tools/run_r04_prof.sh):

    python tools/roofline_table.py <kernel_stats.csv> [--n N --nnz NNZ] > profiles/rNN_kernel_roofline_table.txt

bytes = what the pass is asked to stream (every array it reads or writes once, 8 B per row each; the matrix streams of the product it serves), the fraction is of
the 8 TB/s HBM peak.  Kernels whose bytes depend on run-time arguments this table cannot see (the w of a fused dot may be x itself) carry the lower count and say so."""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--n", type=int, default=512 ** 3)
ap.add_argument("--nnz", type=int, default=7 * 512 ** 3 - 6 * 512 * 512)
a = ap.parse_args()
n, nnz = a.n, a.nnz

RED = {0: (16, "dot (x y)"), 1: (8, "sum of squares (x)"), 2: (8, "nrm1 (x)"), 3: (8, "sum (x)"), 4: (16, "two dots (x y)"),
       5: (48, "CG host-scalar loop: x += a p ; r -= a q ; |r|^2 (p q x r -> x r)"), 6: (56, "the same with <r, r.*dinv> (+ dinv)"),
       7: (24, "y += a x ; |y|^2 (x y -> y)"), 8: (32, "y += a x ; |y|^2, <w,y> (x y w -> y)"), 9: (32, "CG: r -= a q ; |r|^2, <r, r.*dinv> (q r dinv -> r)"),
       10: (24, "CG, uniform diagonal: r -= a q ; |r|^2, <r, c r> (q r -> r)"), 11: (8, "count of entries that differ from a double (x)"),
       12: (56, "BiCGSTAB: x += a phat + w s ; r = s - w t ; |r|^2, <rtld,r> (t s rtld phat x -> x r)"),
       13: (32, "GMRES: one modified Gram-Schmidt step, y -= h x ; <y,w> (x y w -> y)"), 14: (24, "GMRES: the last step, y -= h x ; |y|^2 (x y -> y)")}
EW = {0: (24, "axpy"), 1: (24, "xpay"), 2: (24, "axpyz"), 3: (16, "y = a x"), 4: (24, "z = x .* y"), 5: (24, "z = x ./ y"), 6: (8, "set all"), 7: (16, "abs"), 8: (16, "reciprocal"),
      9: (16, "shift"), 10: (32, "two axpys (x w y -> y)"), 11: (32, "BiCGSTAB direction: p = r + b (p - w v) (v r p -> p)"), 12: (32, "p = x .* d + a p"), 13: (16, "x *= 1/sqrt(s)"), 14: (16, "1/sqrt|x|")}


def bytes_of(name):
    m = re.search(r"spmv_csr_valuerec_march_kernel<2, 2, (\d), (\w+), \d, (\w+), (\w+), (\w+), (\w+)(?:, \w+)?>", name)
    if m:
        dot, ws, gen, box = m.group(1) != "0", m.group(2) == "true", m.group(3) == "true", m.group(4) == "true"
        return ((16 if box else 17) + (8 if ws else 0)) * n, ("headline product, z-marching" + (" (box form: x and y alone)" if box else " (faces by masks)" if not gen else " (general form)") +
                                                           (", fused dots" + (", w a vector of its own" if ws else " (w = x: the diagonal's pair)") if dot else ""))
    m = re.search(r"spmv_csr_valuerec_dom_kernel<256, (\d)", name)
    if m:
        return 17 * n, "headline product (value records, dominant pattern)" + (", fused dots (w = x in CG: no extra stream; w streamed adds 8 B per row)" if m.group(1) != "0" else "")
    m = re.search(r"spmv_csr_rowgather_kernel<\d+, \d+, \d, (\d)", name)
    if m:
        return 12 * nnz + 20 * n + 4, ("CONTRACT FORM: 4 B indices + 8 B values streamed (SURVEY 8d's 12 B per non-zero + 20 B per row), XCD strips" +
                                     (", fused dots (lower count: w may be a stream of its own)" if m.group(1) != "0" else ""))
    m = re.search(r"spmv_csr_pattern7_kernel<256, 2048, (\d)", name)
    if m:
        return 8 * nnz + 17 * n, "product with the values streamed (any 7-point matrix), XCD strips" + (", fused dots (lower count: w may be a stream of its own)" if m.group(1) != "0" else "")
    m = re.search(r"cg_direction_kernel<\w+, \w+, (\d), (\w+)>", name)
    if m:
        xup = m.group(2) == "true"
        jac = int(m.group(1)) == 1
        b = (40 if xup else 24) + (8 if jac else 0)
        return b * n, "CG: " + ("x += a p ; " if xup else "") + "p = M^-1 r + b p" + (" (r p x -> p x)" if xup else " (r p -> p)")
    m = re.search(r"reduce_level1<(\d+),", name)
    if m and int(m.group(1)) in RED:
        b, what = RED[int(m.group(1))]
        return b * n, what
    m = re.search(r"ew_kernel<(\d+),", name)
    if m and int(m.group(1)) in EW:
        b, what = EW[int(m.group(1))]
        return b * n, what
    if "spmv_csr_valuerec_dom_dot4_kernel" in name:
        return 25 * n, "round-3 fused form on the row blocks (A/B legs of the probes)"
    return None, None


rows = []
for r in csv.DictReader(open(a.csv)):
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    b, what = bytes_of(name)
    if b is None:
        continue
    ms = float(r["AverageNs"]) / 1e6
    if ms < 0.05 or b / (ms * 1e-3) > 8e12:
        continue                                    # (launches on small vectors mixed into the average: the reductions' folds, set-up, the 27-point legs)
    rows.append((b / (ms * 1e-3) / 8e12, name, int(r["Calls"]), ms, b, what))
rows.sort(key=lambda t: -t[0])
print(f"# per-kernel roofline table, 512^3 (n = {n:,} rows, {nnz:,} non-zeros), one MI355X: bytes = what the pass is asked to stream (each array once, 8 B per row; the")
print("# matrix streams of the product it serves), fraction of the 8 TB/s HBM peak.  python tools/roofline_table.py <rocprofv3 --kernel-trace --stats CSV of bench.py>")
print(f"# {'kernel':58s} {'calls':>6s} {'avg ms':>8s} {'GB':>8s} {'TB/s':>6s} {'of 8':>6s}  pass")
for frac, name, calls, ms, b, what in rows:
    print(f"  {name[:58]:58s} {calls:6d} {ms:8.4f} {b / 1e9:8.3f} {b / (ms * 1e-3) / 1e12:6.2f} {frac:6.3f}  {what}")
