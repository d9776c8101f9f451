# bench.py over grid sizes: SpMV GFLOP/s (value records and values streamed) and Krylov it/s per size -> gpurun_out/size_sweep.txt
cd $GRAFT_REPO_ROOT
for g in 64 100 128 192 200 256 300 320 384 448 512; do
  timeout 600 python bench.py --grid $g --steps 50 --warmup 5 --preroll 200 --solver-iters 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['krylov']; s=j['values_streamed'] or {}
print(f\"N={$g}^3 n={j['config']['n']:>10d}: SpMV {j['ms_per_step']:.4f} ms {j['value']:8.1f} GFLOP/s ({j['roofline']['kernel']}) | values streamed {s.get('ms_per_step',0):.4f} ms {s.get('value',0):7.1f} GFLOP/s | it/s: CG+Jacobi {k['cg_jacobi']['iters_per_sec']:9.1f}  BiCGSTAB {k['bicgstab_none']['iters_per_sec']:9.1f}  BiCG {k['bicg_none']['iters_per_sec']:9.1f}  GMRES(30) {k['gmres30_none']['iters_per_sec']:8.1f}\")
"
done
