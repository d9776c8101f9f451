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
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, "stdout carries the JSON line alone (what the library and the legs print goes to stderr)"
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
    # the roofline prices the bytes the TIMED kernel moves over its own HIP-event time: a physical fraction; the contract's count is beside it
    assert 0 < r["frac"] <= 1.0 and abs(r["achieved"] - r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) <= (0.01 + 0.00006 / r["kernel_ms"]) * r["achieved"] + 0.1     # (kernel_ms is printed to 4 decimals: 1 % of a 4 us kernel)
    assert r["contract_bytes_per_launch"] == 12 * d["config"]["nnz"] + 20 * d["config"]["n"] + 4 and r["contract_frac"] > 0 and "applies_to" in r
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert d["nontrivial_x"]["value"] > 0 and 0 < d["nontrivial_x"]["frac"] <= 1.0
    vs = d["values_streamed"]
    if r["value_records"]:
        assert vs is not None and vs["roofline"]["value_records"] == 0 and 0 < vs["roofline"]["frac"] <= 1.0
        assert vs["roofline"]["bytes_per_launch"] > r["bytes_per_launch"] and vs["nontrivial_x"]["value"] > 0

    # the contract form (round 5): the same matrix through the kernel that streams the reference's own index[] / value[] arrays, priced on SURVEY 8d's bytes
    cf = d["contract_form"]
    cr = cf["roofline"]
    assert cf["kernel"] == cr["kernel"] == "spmv_csr_rowgather_kernel" and cr["index_codes"] == cr["row_patterns"] == cr["value_records"] == 0
    assert cr["bytes_per_launch"] == cr["contract_bytes_per_launch"] == r["contract_bytes_per_launch"] and 0 < cr["frac"] <= 1.0 and cr["frac"] == cr["contract_frac"]
    assert abs(cr["achieved"] - cr["bytes_per_launch"] / (cr["kernel_ms"] * 1e-3) / 1e9) <= (0.01 + 0.00006 / cr["kernel_ms"]) * cr["achieved"] + 0.1
    assert cf["value"] > 0 and cf["nontrivial_x"]["value"] > 0 and 0 < cf["nontrivial_x"]["frac"] <= 1.0 and cf["xcd_strip_rows"] in (0, 64 * 64)
    assert cf["cg_jacobi"]["iters_per_sec"] > 0 and cf["cg_jacobi"]["loop_bytes_per_iter"] > d["krylov"]["cg_jacobi"]["roofline"]["loop_bytes_per_iter"]

    def fracs(node):
        if isinstance(node, dict):
            for k, v in node.items():
                if k == "frac":
                    yield v
                else:
                    yield from fracs(v)
    assert all(0 < f <= 1.0 for f in fracs(d))
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


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(scaling):
    """the N > 1 launch line of the driver with two ranks sharing this box's GPU (RCCL refuses that: --comm callbacks, a
    bring-up run flagged degraded): partition, halo, the closed-form result check of the (stretched) grid, the collective
    timing and the line's bookkeeping -- weak: 48^3 rows per rank on a 96 x 48 x 48 grid, strong: one 48^3 grid split"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--grid", "48", "--scaling", scaling, "--comm", "callbacks", "--preroll", "5", "--solver-iters", "20",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    planes = 96 if scaling == "weak" else 48
    n = planes * 48 * 48
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["degraded"] is True and d["rccl_ranks"] == 0
    assert d["config"]["n"] == n and d["config"]["nnz"] == 7 * n - 2 * (48 * 48 + 2 * planes * 48)
    assert ("per GPU" in d["config"]["workload"]) == (scaling == "weak")
    assert d["value"] > 0 and d["multi_gpu"] is not None and d["cpu_baseline"] is None
    m = d["multi_gpu"]
    assert m["halo_bytes_per_neighbour"] == 8 * 48 * 48 and m["halo_bytes_per_interior_rank_per_step"] == 2 * 8 * 48 * 48 and m["neighbours_of_rank0"] == 1
    assert m["halo_communicator"] is None and m["folds_per_iteration"]["cg_jacobi"] == 2           # (callbacks: no RCCL communicator of either kind)
    assert m["halo_ms_per_step"] > 0 and m["ms_per_step_no_overlap"] > 0
    assert d["contract_form"]["kernel"] == "spmv_csr_rowgather_kernel" and d["contract_form"]["cg_jacobi"]["iters_per_sec"] > 0
    for name in ("cg_jacobi", "bicgstab_none", "bicg_none", "gmres30_none"):
        assert d["krylov"][name]["iters_per_sec"] > 0
