#!/bin/bash
# Round 5: everything the round's inference numbers come from, one gpurun call (after tools/r5_last2.sh):
#   gpurun --timeout 2400 -- 'bash tools/r5_final.sh'
# the driver-style bench line (+ CPU baseline), the 625-caption shard, the same command under rocprofv3 --kernel-trace
# --stats, FETCH_SIZE / WRITE_SIZE in their own passes (default mode and BASELINE configs[1]: greedy, bf16), the bf16
# greedy line + its kernel stats, the side workloads.
set -u
TAG=r5
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
tail -c 900 "$OUT/${TAG}_bench.json"; echo
timeout 300 python bench.py --captions 625 --steps 20 --warmup 5 --cpu-seconds 0 > "$OUT/${TAG}_bench_625.json" 2> "$OUT/${TAG}_bench_625.err"
tail -c 300 "$OUT/${TAG}_bench_625.json"; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
find "$OUT/${TAG}_kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*kernel_trace.csv" -delete
head -8 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --no-checks --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)" f16x2 5000 "${TAG}"
find "$OUT" -name "*counter_collection.csv" -delete
# ---- BASELINE configs[1]: greedy, MLP mapper, bf16 operands + bf16 KV cache
G="--workload greedy_mlp --gemm-mode bf16 --cpu-seconds 0 --cpu-captions 0 --no-checks"
timeout 300 python bench.py $G --steps 10 --warmup 3 > "$OUT/${TAG}_greedy_bf16_bench.json" 2> "$OUT/${TAG}_greedy_bf16_bench.err"
tail -c 600 "$OUT/${TAG}_greedy_bf16_bench.json"; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_ktg" -- python bench.py $G --no-smi --steps 3 --warmup 1 \
    > "$OUT/${TAG}_greedy_bf16_under_rocprof.json" 2> "$OUT/${TAG}_ktg.err"
find "$OUT/${TAG}_ktg" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_greedy_bf16_kernel_stats.csv" \;
find "$OUT/${TAG}_ktg" -name "*kernel_trace.csv" -delete
head -8 "$OUT/${TAG}_greedy_bf16_kernel_stats.csv" | cut -c1-170
mkdir -p "$OUT/g16"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/g16/${TAG}_pmc_$c" -- python bench.py $G --no-smi --steps 1 --warmup 0 \
        > "$OUT/g16/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT/g16" "$OUT/${TAG}_greedy_bf16_pmc_traffic.json" \
    "python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 1 --warmup 0 (BASELINE configs[1]: 5000 captions, greedy, T=67)" bf16 5000 "${TAG}"
find "$OUT" -name "*counter_collection.csv" -delete
rm -rf "$OUT/g16"
bash tools/side_workloads.sh "$TAG"
