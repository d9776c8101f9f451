# round 3's committed profile set: kernel-trace stats + PMC passes of bench.py, traffic json of both timed kernels, the bench line
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40"
PROF_PASS_TIMEOUT=400 bash tools/prof.sh r3final python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --preroll 100 --no-cpu-baseline --solver-iters 40 > gpurun_out/r3final_summary.txt 2>&1
python tools/traffic_json.py gpurun_out/r3final "spmv_csr_valuerec_dom_kernel<256, 0>" gpurun_out/r3final_traffic.json --patterns 27 --values 1 --command "$CMD" > gpurun_out/r3final_traffic.log 2>&1
python tools/traffic_json.py gpurun_out/r3final "spmv_csr_pattern7_kernel<256, 2048, 0>" gpurun_out/r3final_traffic_streamed.json --patterns 27 --command "$CMD" > gpurun_out/r3final_traffic_streamed.log 2>&1
tail -3 gpurun_out/r3final_traffic.log
cp gpurun_out/r3final/trace/*kernel_stats.csv gpurun_out/r3final_kernel_stats.csv 2>/dev/null || find gpurun_out/r3final/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3final_kernel_stats.csv \;
timeout 900 python bench.py > gpurun_out/r3final_bench.json 2> gpurun_out/r3final_bench.err
tail -c 400 gpurun_out/r3final_bench.json
