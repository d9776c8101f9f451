"""bench.py prints ONE JSON line with the fields the driver reads (small grid, seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--grid", "64",
                        "--preroll", "5", "--solver-iters", "20"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "GFLOP/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["preroll"] == 5 and d["degraded"] is False and d["rccl_ranks"] is None
    for name in ("cg_jacobi", "bicgstab_none", "bicg_none", "gmres30_none"):
        k = d["krylov"][name]
        assert k["iters_per_sec"] > 0 and 0 < k["itime_s"] <= k["lis_solve_wall_s"] + 1e-3
        assert abs(k["iters_per_sec"] - k["iters_timed"] / k["itime_s"]) < 0.01 * k["iters_per_sec"]     # iter / itime (lis_solver.c:902-908)
        kr = k["roofline"]
        assert 0 < kr["frac"] <= 1.0 and kr["loop_bytes_per_iter"] <= kr["contract_bytes_per_iter"]
