# the round's committed profile set: kernel-trace stats + PMC passes of bench.py, traffic json, bench line
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40"
PROF_PASS_TIMEOUT=400 bash tools/prof.sh r2final python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r2final_summary.txt 2>&1
python tools/traffic_json.py gpurun_out/r2final "spmv_csr_valuerec_pair_kernel<256, 1>" gpurun_out/r2final_traffic.json --patterns 27 --values 1 --command "$CMD" > gpurun_out/r2final_traffic.log 2>&1
python tools/traffic_json.py gpurun_out/r2final "spmv_csr_pattern7_kernel<256, 2048, 0>" gpurun_out/r2final_traffic_streamed.json --patterns 27 --command "$CMD" > gpurun_out/r2final_traffic_streamed.log 2>&1
tail -3 gpurun_out/r2final_traffic.log
timeout 900 python bench.py > gpurun_out/r2final_bench.json 2> gpurun_out/r2final_bench.err
tail -c 400 gpurun_out/r2final_bench.json
