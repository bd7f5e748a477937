#!/bin/bash
# round-2 probe 2: f16x2 vs bf16x3 GEMM micro-benchmarks, bench lines in both modes, full GPU test suite
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
for m in f16x2 bf16x3; do
  CAPDEC_GEMM_MODE=$m CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 timeout 300 python tools/gemm_bench.py 25000 3125 > "$OUT/r2_gemm_$m.json" 2> "$OUT/r2_gemm_$m.err"
  cat "$OUT/r2_gemm_$m.json"
done
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm or f16x2 or logits or decode_tiny or decode_small" > "$OUT/r2_pytest_core.txt" 2>&1
tail -5 "$OUT/r2_pytest_core.txt"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_bench_f16x2.json" 2> "$OUT/r2_bench_f16x2.err"
cut -c1-300 "$OUT/r2_bench_f16x2.json"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --gemm-mode bf16x3 > "$OUT/r2_bench_bf16x3.json" 2> "$OUT/r2_bench_bf16x3.err"
cut -c1-300 "$OUT/r2_bench_bf16x3.json"
timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_bench_f16x2_625.json" 2> "$OUT/r2_bench_f16x2_625.err"
cut -c1-300 "$OUT/r2_bench_f16x2_625.json"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r2_pytest_all.txt" 2>&1
tail -8 "$OUT/r2_pytest_all.txt"
