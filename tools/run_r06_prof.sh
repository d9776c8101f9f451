#!/bin/bash
# The round-6 evidence run (one gpurun call): kernel stats + PMC passes of bench.py, the traffic JSONs of the three forms of the headline product, the native
# format kernels at 512^3 and 256^3 under the profiler, the format / irregular / size sweeps, and the bench line itself.  Everything lands under gpurun_out/r06_*;
# the summaries worth keeping are copied to profiles/ by hand.
set -u
cd $GRAFT_REPO_ROOT
export PROF_PMC_ONLY="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum|TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum|TCC_HIT_sum TCC_MISS_sum"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --no-live-traffic --no-extras --solver-iters 40"
tools/prof.sh r06_prof $CMD > gpurun_out/r06_prof_summary.txt 2>&1
python tools/traffic_json.py gpurun_out/r06_prof "spmv_csr_rowgather_kernel<256, 2048, 7, 0>" gpurun_out/r06_spmv512_traffic_contract_form.json --coded 0 --command "$CMD" > /dev/null 2>&1
python tools/traffic_json.py gpurun_out/r06_prof "spmv_csr_valuerec_march_kernel<2, 2, 0, false" gpurun_out/r06_spmv512_traffic.json --values 1 --box 1 --patterns 1 --command "$CMD" > /dev/null 2>&1
python tools/traffic_json.py gpurun_out/r06_prof "spmv_csr_pattern7_kernel<256, 2048, 0>" gpurun_out/r06_spmv512_traffic_values_streamed.json --patterns 1 --command "$CMD" > /dev/null 2>&1
cp gpurun_out/r06_prof/trace/*/*kernel_stats.csv gpurun_out/r06_bench_spmv512_kernel_stats.csv 2>/dev/null || find gpurun_out/r06_prof/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_bench_spmv512_kernel_stats.csv \;
tools/prof_formats.sh r06_formats_512 512 csr,ell,dia,bsr > gpurun_out/r06_formats_512.txt 2>&1
tools/prof_formats.sh r06_formats_256 256 csr,ell,dia,bsr > gpurun_out/r06_formats_256.txt 2>&1
{ echo "== tests/perf/format_sweep.py (256^3, default forms)"; python tests/perf/format_sweep.py 2>&1 | tail -40;
  echo "== tests/perf/irregular_sweep.py (default: long-row tree on)"; python tests/perf/irregular_sweep.py 2>&1 | grep -E "^fem3|^zipf" ;
  echo "== LIS_AMD_LONG_ROW_CHAIN=1 zipf"; IRREG_ONLY=zipf LIS_AMD_LONG_ROW_CHAIN=1 python tests/perf/irregular_sweep.py 2>&1 | grep -E "^zipf" ; } > gpurun_out/r06_format_irregular_sweeps.txt 2>&1
{ echo "== tests/perf/irregular_sweep.py mesh, 2 M nodes (short rows through block-local columns: round 6) and with LIS_AMD_NO_LOCAL_SHORT_ROWS=1 (the rule of rounds 2-5)";
  IRREG_ONLY=mesh IRREG_MESH_NODES=2000000 python tests/perf/irregular_sweep.py 2>&1 | grep -E "^mesh";
  LIS_AMD_NO_LOCAL_SHORT_ROWS=1 IRREG_ONLY=mesh IRREG_MESH_NODES=2000000 python tests/perf/irregular_sweep.py 2>&1 | grep -E "^mesh";
  echo "== 8 M nodes: as it comes / renumbered at plan time (LIS_AMD_REORDER_AFTER=0) / rounds 2-5 (no short-row lists, no renumbering)";
  IRREG_ONLY=mesh IRREG_MESH_NODES=8000000 python tests/perf/irregular_sweep.py --gmres-iters 100 2>&1 | grep -E "^mesh";
  LIS_AMD_REORDER_TRACE=1 LIS_AMD_REORDER_AFTER=0 IRREG_ONLY=mesh IRREG_MESH_NODES=8000000 python tests/perf/irregular_sweep.py --gmres-iters 100 2>&1 | grep -E "^mesh|reorder:";
  LIS_AMD_NO_LOCAL_SHORT_ROWS=1 LIS_AMD_NO_REORDER=1 IRREG_ONLY=mesh IRREG_MESH_NODES=8000000 python tests/perf/irregular_sweep.py --gmres-iters 100 2>&1 | grep -E "^mesh";
  echo "== tools/local_short_rows_probe.py: numberings with less locality (4 M nodes, 8^3 and 4^3 cells)";
  python tools/local_short_rows_probe.py mesh 4000000 8 2>&1 | grep -E "^G=|P A P|renumbered"; python tools/local_short_rows_probe.py mesh 4000000 4 2>&1 | grep -E "^G=|P A P|renumbered";
  echo "== tools/yardstick_sweep.py"; python tools/yardstick_sweep.py 2>&1 | grep yardstick; } > gpurun_out/r06_unstructured_mesh_and_yardstick.txt 2>&1
bash tools/size_sweep.sh > gpurun_out/r06_size_sweep.txt 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
echo done; tail -3 gpurun_out/r06_size_sweep.txt; python tools/show_bench.py gpurun_out/r06_bench_line.json 1 | head -8
