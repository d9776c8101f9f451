#!/bin/bash
# tools/scale8.sh -- the scaling curve of the hot path on ONE node: bench.py at N = 1, 2, 4, 8 GPUs, weak (512^3 rows per GPU) and strong (the one 512^3 grid
# split into row blocks of whole planes), one rank per GPU over RCCL, exactly the launch line the driver uses.  Every run prints bench.py's one JSON line; they are
# kept in $OUT/scale8_<weak|strong>.jsonl and summarised against the prediction committed in profiles/r05_scale_prediction.json (tools/scale8_summary.py).
#
#   tools/scale8.sh [max_gpus=8] [steps=50] [solver_iters=300]
#
# A run that could not form its RCCL communicator falls back to host callbacks, marks its line "degraded": true and exits 3: the summary flags it, it is never
# counted as a measurement.  LIS_AMD_COMM_TIMEOUT (default 300 s) turns a rank that never joins into an abort with a message instead of a hang.
set -u
cd "$(dirname "$0")/.."
MAXN=${1:-8}; STEPS=${2:-50}; ITERS=${3:-300}
OUT=${OUT:-gpurun_out}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LIS_AMD_COMM_TIMEOUT=${LIS_AMD_COMM_TIMEOUT:-300}
for MODE in weak strong; do
  : > "$OUT/scale8_$MODE.jsonl"
  for N in 1 2 4 8; do
    [ "$N" -gt "$MAXN" ] && continue
    echo "== $MODE scaling, $N GPU(s)" >&2
    # the driver's own entry point: `python bench.py --gpus N` starts its N ranks itself (one per GPU, torch.distributed.run on 127.0.0.1)
    timeout 1800 python bench.py --gpus "$N" --steps "$STEPS" --warmup 5 --scaling "$MODE" --solver-iters "$ITERS" --no-extras --no-live-traffic --no-cpu-baseline \
      2> "$OUT/scale8_${MODE}_$N.err" | grep '^{' >> "$OUT/scale8_$MODE.jsonl"
    echo "   exit ${PIPESTATUS[0]}" >&2
  done
done
python tools/scale8_summary.py "$OUT/scale8_weak.jsonl" "$OUT/scale8_strong.jsonl" | tee "$OUT/scale8_summary.json"
