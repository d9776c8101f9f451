"""The predicted scaling curve of bench.py --gpus N (N = 1, 2, 4, 8; weak and strong), built from what ONE GPU can measure plus a stated model of the node's links:

    python tools/scale_predict.py slabs.jsonl > profiles/r05_scale_prediction.json

slabs.jsonl: bench.py lines of `--planes L` runs on one GPU for L = 512, 256, 128, 64 (512 x 512 x L rows: the slab one rank of the strong-scaling job owns at
N = 1, 2, 4, 8; the weak-scaling job keeps the L = 512 slab on every rank).  From each line: the product's kernel time, CG + Jacobi and BiCGSTAB seconds per iteration.

Model (every constant is here, none is fitted to a multi-GPU run -- there has not been one):
  * halo: one 512 x 512 plane of doubles (2 MiB) out and in per neighbour, both neighbours at once over different xGMI links.  A link moves ~64 GB/s per direction at
    its 153.6 GB/s bidirectional rating; RCCL send/recv of 2 MiB is taken at 40-60 GB/s plus 15-25 us of launch / handshake latency: t_halo = 50 us (band 40-80).
  * the exchange overlaps the interior rows (all planes but the two boundary ones); exposed = max(0, t_halo - t_interior); the two boundary launches add ~2 x 5 us.
  * a fold = ncclAllGather of <= 32 B + the rank-order add inside the next scalar step, no host round trip: 15 / 20 / 30 us at 2 / 4 / 8 ranks (band x 0.7 .. x 1.5).
    CG + Jacobi has 2 folds per iteration, BiCGSTAB 4 (bench.py multi_gpu.folds_per_iteration).
  * everything else per rank is what the slab measured on one GPU.
Output: {"weak": {"1": {...}, ...}, "strong": {...}, "model": {...}, "inputs": {...}}; value = whole-job SpMV GFLOP/s, *_iters_per_sec whole-job iterations/s."""
import json
import sys

lines = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
slab = {}
for l in lines:
    n = l["config"]["n"]
    L = n // (512 * 512)
    k = l["krylov"]
    slab[L] = {"spmv_ms": l["roofline"]["kernel_ms"], "step_ms": l["ms_per_step"], "nnz": l["config"]["nnz"],
               "cg_ms": 1e3 / k["cg_jacobi"]["iters_per_sec"], "bicgstab_ms": 1e3 / k["bicgstab_none"]["iters_per_sec"],
               "contract_ms": (l.get("contract_form") or {}).get("kernel_ms")}
HALO = {"mid": 0.050, "lo": 0.040, "hi": 0.080}                       # ms
FOLD = {2: 0.015, 4: 0.020, 8: 0.030}                                 # ms per fold
BND_LAUNCH = 0.010                                                    # ms: two more launches per product in a multi-rank job
FOLDS = {"cg": 2, "bicgstab": 4}
PRODUCTS = {"cg": 1, "bicgstab": 2}


def point(L_rank, N, nnz_global, halo, fold_scale):
    s = slab[L_rank]
    if N == 1:
        return {"value": round(2e-6 * nnz_global / s["step_ms"], 1), "ms_per_step": round(s["step_ms"], 4),
                "cg_jacobi_iters_per_sec": round(1e3 / s["cg_ms"], 1), "bicgstab_iters_per_sec": round(1e3 / s["bicgstab_ms"], 1)}
    t_int = s["spmv_ms"] * (L_rank - 2) / L_rank
    exposed = max(0.0, halo - t_int)
    step = s["step_ms"] + exposed + BND_LAUNCH
    f = FOLD[N] * fold_scale
    cg = s["cg_ms"] + PRODUCTS["cg"] * (exposed + BND_LAUNCH) + FOLDS["cg"] * f
    bi = s["bicgstab_ms"] + PRODUCTS["bicgstab"] * (exposed + BND_LAUNCH) + FOLDS["bicgstab"] * f
    return {"value": round(2e-6 * nnz_global / step, 1), "ms_per_step": round(step, 4), "exposed_halo_ms": round(exposed, 4),
            "cg_jacobi_iters_per_sec": round(1e3 / cg, 1), "bicgstab_iters_per_sec": round(1e3 / bi, 1)}


def nnz_of(L):                                                        # 7-point stencil on an L x 512 x 512 grid
    n = L * 512 * 512
    return 7 * n - 2 * (512 * 512 + 2 * L * 512)


out = {"weak": {}, "strong": {}}
for N in (1, 2, 4, 8):
    for mode, L_rank, L_glob in (("weak", 512, 512 * N), ("strong", 512 // N, 512)):
        if L_rank not in slab:
            continue
        mid = point(L_rank, N, nnz_of(L_glob), HALO["mid"], 1.0)
        lo = point(L_rank, N, nnz_of(L_glob), HALO["hi"], 1.5)
        hi = point(L_rank, N, nnz_of(L_glob), HALO["lo"], 0.7)
        mid["band"] = {k: [lo[k], hi[k]] for k in ("value", "cg_jacobi_iters_per_sec", "bicgstab_iters_per_sec")}
        base = point(512, 1, nnz_of(512), 0, 1)
        mid["efficiency"] = round(mid["value"] / (N * base["value"]), 3)
        out[mode][str(N)] = mid
out["model"] = {"halo_ms": HALO, "fold_ms": FOLD, "boundary_launch_ms": BND_LAUNCH, "folds_per_iteration": FOLDS, "products_per_iteration": PRODUCTS,
                "note": "constants stated, not fitted: no multi-GPU run of this path exists yet (SCALE_r01..r04 skipped for lack of an 8-GPU node)"}
out["inputs"] = {str(k): v for k, v in sorted(slab.items())}
print(json.dumps(out, indent=1))
