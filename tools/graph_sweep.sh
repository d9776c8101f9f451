# hipGraph replay of the Krylov batches, A/B over sizes (bench.py's solver lines; LIS_AMD_GRAPHS=0 plain launches, =1 replay at any size)
cd $GRAFT_REPO_ROOT
for g in ${GRIDS:-32 64 100 128 160 200 256 320}; do
  for m in 0 1 0 1; do
    LIS_AMD_GRAPHS=$m timeout 600 python bench.py --grid $g --steps 20 --warmup 5 --preroll 50 --solver-iters 400 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['krylov']
print(f\"N={$g}^3 graphs={$m}: it/s CG+Jacobi {k['cg_jacobi']['iters_per_sec']:9.1f}  BiCGSTAB {k['bicgstab_none']['iters_per_sec']:9.1f}  BiCG {k['bicg_none']['iters_per_sec']:9.1f}  GMRES(30) {k['gmres30_none']['iters_per_sec']:8.1f}\")
"
  done
done
