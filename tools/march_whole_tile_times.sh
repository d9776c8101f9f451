# Timings of the three marching kernels at whole-tile sizes (HEAD; numbers to compare: profiles/EXPERIMENTS.md, "partial-tile tests folded away")
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 50 --warmup 5 --preroll 200 --no-cpu-baseline --no-extras --no-live-traffic --no-solvers"
one() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   7-point", d["config"]["n"], "rows:", d["roofline"]["kernel_ms"], "ms", d["roofline"]["kernel"])
P
}
for r in 1 2; do
  python tools/stencil27_probe.py 256 2>&1 | grep "variant 0x0" | tail -1
  python tests/perf/bsr_sweep.py 256 2>&1 | grep "bsr 2x2 team 0"
  for g in 128 256 512; do $B --grid $g > /tmp/b.json 2>/dev/null; one /tmp/b.json; done
done
