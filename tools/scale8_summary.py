"""Summarise the lines tools/scale8.sh collected: one SCALE-shaped JSON object per scaling mode on stdout, a table on stderr, each point beside the prediction of
profiles/r05_scale_prediction.json (tools/scale_predict.py) when that file is there.

    python tools/scale8_summary.py scale8_weak.jsonl scale8_strong.jsonl

Efficiency: weak = value(N) / (N * value(1)); strong = value(N) / (N * value(1)) too (the job is the same 512^3 product: value is whole-job GFLOP/s)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pred = None
try:
    pred = json.load(open(os.path.join(ROOT, "profiles", "r05_scale_prediction.json")))
except OSError:
    pass
out = {}
for path in sys.argv[1:]:
    lines = [json.loads(l) for l in open(path) if l.startswith("{")]
    if not lines:
        continue
    mode = lines[-1]["scaling"] if len(lines) > 1 else ("strong" if "strong" in path else "weak")
    base = next((l for l in lines if l["n_gpus"] == 1), None)
    pts = []
    for l in lines:
        n = l["n_gpus"]
        k = l.get("krylov") or {}
        m = l.get("multi_gpu") or {}
        p = {"n_gpus": n, "value": l["value"], "unit": l["unit"], "ms_per_step": l["ms_per_step"], "rccl_ranks": l.get("rccl_ranks"), "degraded": l.get("degraded"),
             "halo_bytes_per_interior_rank_per_step": m.get("halo_bytes_per_interior_rank_per_step"), "halo_ms_alone": m.get("halo_ms_per_step"),
             "ms_per_step_no_overlap": m.get("ms_per_step_no_overlap"), "halo_communicator": m.get("halo_communicator"),
             "cg_jacobi_iters_per_sec": (k.get("cg_jacobi") or {}).get("iters_per_sec"), "bicgstab_iters_per_sec": (k.get("bicgstab_none") or {}).get("iters_per_sec"),
             "roofline_frac": (l.get("roofline") or {}).get("frac"),
             "structured_fast_path_value": (l.get("structured_fast_path") or {}).get("value"),
             "structured_fast_path_cg_jacobi_iters_per_sec": (((l.get("structured_fast_path") or {}).get("krylov") or {}).get("cg_jacobi") or {}).get("iters_per_sec"),
             "efficiency": round(l["value"] / (n * base["value"]), 4) if base else None}
        if pred and mode in pred and str(n) in pred[mode]:
            q = pred[mode][str(n)]
            p["predicted"] = q
            # (the prediction of round 5 was made for the plan's default form of this matrix -- round 6's `structured_fast_path`; the headline is the reference layout)
            for key, mine in (("value", "structured_fast_path_value"), ("cg_jacobi_iters_per_sec", "structured_fast_path_cg_jacobi_iters_per_sec")):
                if q.get(key) and p.get(mine):
                    p[f"{mine}_over_predicted"] = round(p[mine] / q[key], 3)
        pts.append(p)
        print(f"{mode:6s} N={n}  {l['value']:10.1f} GFLOP/s  eff {p['efficiency']}  cg {p['cg_jacobi_iters_per_sec']} it/s  bicgstab {p['bicgstab_iters_per_sec']} it/s  "
              f"rccl_ranks {p['rccl_ranks']} degraded {p['degraded']}" + (f"  predicted {p['predicted'].get('value')}" if "predicted" in p else ""), file=sys.stderr)
    out[mode] = {"metric": lines[0]["metric"], "scaling": mode, "points": pts}
print(json.dumps(out))
