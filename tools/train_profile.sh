#!/bin/bash
# Per-kernel table of the train step with a frozen GPT-2 (bench.py --workload train_step) on an MI355X box:
#   gpurun --timeout 300 -- 'bash tools/train_profile.sh r5'   ->  gpurun_out/<tag>_train_kernels.txt (+ the bench line)
# rocprofv3 --kernel-trace of a short run, summarised per (kernel, grid) by tools/trace_summary.py: which of the ~500
# launches of a step the time goes to (the hipEvent families of the bench line only cover the launchers that carry a
# ProfScope: GEMMs, LayerNorm, attention forward).
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
timeout 120 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_train_step_bench.json" 2> "$OUT/${TAG}_train_step_bench.err"
timeout 150 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_kt_train" -- python bench.py --workload train_step --steps 3 --warmup 1 --cpu-seconds 0 \
    > "$OUT/${TAG}_kt_train.json" 2> "$OUT/${TAG}_kt_train.err"
python tools/trace_summary.py "$OUT/${TAG}_kt_train" "$OUT/${TAG}_train_kernels.txt" --title "bench.py --workload train_step --steps 3 --warmup 1 under rocprofv3 --kernel-trace (4 steps + weight transposes of the first)"
find "$OUT/${TAG}_kt_train" -name "*.csv" -delete
tail -c 700 "$OUT/${TAG}_train_step_bench.json"; echo
head -40 "$OUT/${TAG}_train_kernels.txt"
