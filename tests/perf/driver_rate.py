"""The rate the reference's UNCHANGED spmvtest3 binary reports (its own host clock around 100 products) in the three coherence modes -- a wall-clock figure that
belongs here and not in the correctness suite (a busy box moves it by 2x):

    python tests/perf/driver_rate.py [N=200] [iters=100]

Expected on a quiet MI355X box at 200^3: default (page protection) and resident 1.5-1.6 TFLOP/s -- 70 us per call by the driver's clock, mostly host time around a
30 us kernel --, eager (copies on every call) ~46 GFLOP/s, a PCIe figure.  Exit code 1 when the default mode is below half of the resident one or eager is not far
below both (the properties that do not depend on the box's load)."""
import os
import re
import subprocess
import sys

DRV = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "drivers", "spmvtest3_amd")
N = sys.argv[1] if len(sys.argv) > 1 else "200"
iters = sys.argv[2] if len(sys.argv) > 2 else "100"
rate = {}
for mode, env in (("default", {}), ("eager", {"LIS_AMD_COHERENCE": "eager"}), ("resident", {"LIS_AMD_RESIDENCY": "resident"})):
    out = subprocess.run([DRV, N, N, N, iters, "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1", **env), check=True).stdout
    m = re.search(r"computation = (\S+) sec, (\S+) MFLOPS, 2-norm = (\S+)", out)
    rate[mode] = float(m.group(2)) * 1e-3
    print(f"{mode:9s} {rate[mode]:9.1f} GFLOP/s   2-norm {m.group(3)}", flush=True)
ok = rate["default"] >= 0.5 * rate["resident"] and rate["eager"] < 0.2 * rate["default"]
print("ok" if ok else "UNEXPECTED: see the docstring")
sys.exit(0 if ok else 1)
