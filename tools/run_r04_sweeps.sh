# round 4's sweeps beside the bench: formats at 256^3 (constant coefficients and values streamed), the irregular matrices, BSR with long block rows, the Queen-class product
cd $GRAFT_REPO_ROOT
F="^initial\|^precision\|^linear\|^precond\|^converg\|^matrix st\|^$"
{
echo "# round 4, final tree, one MI355X.  python tests/perf/format_sweep.py 256   (constant-coefficient stencil: every format's row form runs the dominant-pattern value-record kernel; BSR 2 x 2: the row form through the wide value records)"
python tests/perf/format_sweep.py 256 2>&1 | grep -v "$F"
echo "# LIS_AMD_NO_VALUE_RECORDS=1 python tests/perf/format_sweep.py 256   (values streamed: CSR / CSC / JAD spmv_csr_pattern7_kernel with XCD strips, ELL / DIA / BSR their native kernels)"
LIS_AMD_NO_VALUE_RECORDS=1 python tests/perf/format_sweep.py 256 2>&1 | grep -v "$F"
echo "# python tests/perf/irregular_sweep.py   (fem3: block-local columns kernel, positions in registers, four workgroups per CU; zipf: products kernel with the fed chain)"
python tests/perf/irregular_sweep.py 80 2>&1 | grep -v "$F"
echo "# IRREG_ROUND3=1 IRREG_ONLY=fem3 python tests/perf/irregular_sweep.py 80 --gmres-iters 0   (the round-3 form of the block-local kernel, same box)"
IRREG_ROUND3=1 IRREG_ONLY=fem3 python tests/perf/irregular_sweep.py 80 --gmres-iters 0 2>&1 | grep "SpMV"
echo "# python tests/perf/bsr_sweep.py --fem {110 2, 100 3, 96 4}"
for a in "110 2" "100 3" "96 4"; do python tests/perf/bsr_sweep.py --fem $a 2>&1 | grep "bsr "; done
echo "# python tools/queen_probe.py 200   /   QUEEN_ROUND3=1 python tools/queen_probe.py 200   (Queen-class product: round 4 / round 3 form, same box)"
python tools/queen_probe.py 200 2>&1 | tail -1
QUEEN_ROUND3=1 python tools/queen_probe.py 200 2>&1 | tail -1
echo "# DOM_FORMS=none python tools/dom_probe.py 512 2   /   NO_XCD_STRIPS=1 ...   (values-streamed 7-point product, non-trivial x: XCD strips on / off, same box)"
DOM_FORMS=none python tools/dom_probe.py 512 2 2>&1 | grep "values streamed"
NO_XCD_STRIPS=1 DOM_FORMS=none python tools/dom_probe.py 512 2 2>&1 | grep "values streamed"
} > gpurun_out/r04_sweeps.txt 2>&1
tail -40 gpurun_out/r04_sweeps.txt
