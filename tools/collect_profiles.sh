#!/bin/bash
# Reproduce everything under profiles/ on an MI355X box (one gpurun call):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r1'
# then copy gpurun_out/<tag>_* into profiles/.  Steps: the default bench line (with the CPU baseline), the same command
# under rocprofv3 --kernel-trace --stats (kernel averages must agree with the hipEvent averages of the bench line),
# and the FETCH_SIZE / WRITE_SIZE counters in their own passes (never combined with tracing), summarised per kernel.
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- python bench.py --cpu-seconds 0 \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
find "$OUT/${TAG}_kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*domain_stats.csv" -exec cp {} "$OUT/${TAG}_bench_domain_stats.csv" \;
find "$OUT/${TAG}_kt" -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)"
find "$OUT" -name "*counter_collection.csv" -delete
tail -c 600 "$OUT/${TAG}_bench.json"; echo
head -5 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-160
