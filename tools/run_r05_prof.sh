# round 5's committed profile set: kernel-trace stats + PMC passes of bench.py, traffic json of the three timed kernels (headline, values streamed, CONTRACT FORM),
# the roofline table, the bench line.  Run on the GPU box:  bash tools/run_r05_prof.sh
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --no-extras --no-live-traffic --solver-iters 40"
PROF_PASS_TIMEOUT=400 bash tools/prof.sh r5final python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --no-extras --no-live-traffic --solver-iters 40 > gpurun_out/r5final_summary.txt 2>&1
python tools/traffic_json.py gpurun_out/r5final "spmv_csr_valuerec_march_kernel<2, 2, 0, false, 1, false, true, false, false, false>" gpurun_out/r5final_traffic.json --patterns 27 --values 1 --box 1 --command "$CMD" > gpurun_out/r5final_traffic.log 2>&1
python tools/traffic_json.py gpurun_out/r5final "spmv_csr_pattern7_kernel<256, 2048, 0>" gpurun_out/r5final_traffic_streamed.json --patterns 27 --command "$CMD" > gpurun_out/r5final_traffic_streamed.log 2>&1
python tools/traffic_json.py gpurun_out/r5final "spmv_csr_rowgather_kernel<256, 2048, 7, 0>" gpurun_out/r5final_traffic_contract_form.json --coded 0 --command "$CMD" > gpurun_out/r5final_traffic_contract_form.log 2>&1
tail -3 gpurun_out/r5final_traffic_contract_form.log
find gpurun_out/r5final/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r5final_kernel_stats.csv \;
python tools/roofline_table.py gpurun_out/r5final_kernel_stats.csv > gpurun_out/r5final_roofline_table.txt
timeout 900 python bench.py > gpurun_out/r5final_bench.json 2> gpurun_out/r5final_bench.err
tail -c 300 gpurun_out/r5final_bench.json
