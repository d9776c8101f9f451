# end of round 5 (after the PT instantiations): the size sweep and the 256^3 format / BSR sweeps again -> gpurun_out/r05_late_*.txt
cd $GRAFT_REPO_ROOT
F="^initial\|^precision\|^linear\|^precond\|^converg\|^matrix st\|^$\|^liblis_amd"
bash tools/size_sweep.sh > gpurun_out/r05_late_size_sweep.txt 2>&1
{
echo "# python tests/perf/format_sweep.py 256"
python tests/perf/format_sweep.py 256 2>&1 | grep -v "$F"
echo "# python tests/perf/bsr_sweep.py 256"
python tests/perf/bsr_sweep.py 256 2>&1 | grep "bsr "
echo "# python tools/stencil27_probe.py 256 (constant coefficients, the marching kernel)"
python tools/stencil27_probe.py 256 2>&1 | grep "variant 0x0" | tail -1
} > gpurun_out/r05_late_formats.txt 2>&1
tail -30 gpurun_out/r05_late_size_sweep.txt | cut -c1-200
cat gpurun_out/r05_late_formats.txt | cut -c1-160
