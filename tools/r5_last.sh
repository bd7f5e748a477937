#!/bin/bash
# Round 5, last call: the whole GPU suite on the final tree, smoke(), then the lines that changed after tools/r5_final.sh
# (bf16 mode with K / V through the qkv epilogue; train step) and the headline once more
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/r5_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/r5_pytest_gpu.txt"; tail -34 "$OUT/r5_pytest_gpu.txt" | cut -c1-160
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/r5_bench_final.json" 2> "$OUT/r5_bench_final.err"; tail -c 500 "$OUT/r5_bench_final.json"; echo
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
$B --workload greedy_mlp --gemm-mode bf16 --steps 10 --warmup 3 > "$OUT/r5_greedy_bf16_bench_final.json" 2>/dev/null; tail -c 300 "$OUT/r5_greedy_bf16_bench_final.json"; echo
$B --gemm-mode bf16 --steps 3 --warmup 1 > "$OUT/r5_beam_bf16_bench_final.json" 2>/dev/null; tail -c 300 "$OUT/r5_beam_bf16_bench_final.json"; echo
