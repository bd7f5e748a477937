#!/bin/bash
# rocprofv3 --kernel-trace --stats of the train step (both scopes) and of BASELINE configs[1] on the final tree
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
for sc in prefix full; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r5_ktt_$sc" -- python bench.py --workload train_step --train-scope $sc --steps 10 --warmup 2 --cpu-seconds 0 \
        > "$OUT/r5_train_${sc}_under_rocprof.json" 2> "$OUT/r5_ktt_$sc.err"
    find "$OUT/r5_ktt_$sc" -name "*kernel_stats.csv" -exec cp {} "$OUT/r5_train_${sc}_kernel_stats.csv" \;
    find "$OUT/r5_ktt_$sc" -name "*kernel_trace.csv" -delete
    head -7 "$OUT/r5_train_${sc}_kernel_stats.csv" | cut -c1-160
done
G="--workload greedy_mlp --gemm-mode bf16 --cpu-seconds 0 --cpu-captions 0 --no-checks"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r5_ktg2" -- python bench.py $G --no-smi --steps 3 --warmup 1 \
    > "$OUT/r5_greedy_bf16_under_rocprof.json" 2> "$OUT/r5_ktg2.err"
find "$OUT/r5_ktg2" -name "*kernel_stats.csv" -exec cp {} "$OUT/r5_greedy_bf16_kernel_stats.csv" \;
find "$OUT/r5_ktg2" -name "*kernel_trace.csv" -delete
head -6 "$OUT/r5_greedy_bf16_kernel_stats.csv" | cut -c1-160
timeout 300 python bench.py $G --steps 10 --warmup 3 > "$OUT/r5_greedy_bf16_bench.json" 2>/dev/null; tail -c 200 "$OUT/r5_greedy_bf16_bench.json"; echo
