# bench.py over grid sizes: SpMV GFLOP/s (value records and values streamed) and Krylov it/s per size -> gpurun_out/size_sweep.txt
cd $GRAFT_REPO_ROOT
for g in 64 100 128 192 200 256 300 320 384 448 512; do
  timeout 600 python bench.py --grid $g --steps 50 --warmup 5 --preroll 200 --solver-iters 300 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['krylov']; f=j['structured_fast_path']; s=f.get('values_streamed') or {}; fk=f['krylov']
print(f\"N={$g}^3 n={j['config']['n']:>10d}: reference layout {j['ms_per_step']:.4f} ms {j['value']:8.1f} GFLOP/s frac {j['roofline']['frac']:.3f} CG+Jacobi {k['cg_jacobi']['iters_per_sec']:8.1f} it/s | default form {f['ms_per_step']:.4f} ms {f['value']:8.1f} GFLOP/s ({f['kernel']}) | values streamed {s.get('ms_per_step',0):.4f} ms {s.get('value',0):7.1f} GFLOP/s | it/s (default form): CG+Jacobi {fk['cg_jacobi']['iters_per_sec']:9.1f}  BiCGSTAB {fk['bicgstab_none']['iters_per_sec']:9.1f}  BiCG {fk['bicg_none']['iters_per_sec']:9.1f}  GMRES(30) {fk['gmres30_none']['iters_per_sec']:8.1f}\")
"
done
