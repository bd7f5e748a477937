#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
for mid in 0 1; do
  CAPDEC_SPLITK_MID=$mid timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_bench_625_mid$mid.json" 2>/dev/null
  python -c "
import json; r=json.load(open('$OUT/r2_bench_625_mid$mid.json')); print('625 mid$mid:', r['value'], {k:v['avg_ms'] for k,v in r['kernels'].items() if 'gemm' in k})"
done
CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1 timeout 200 python tools/gemm_bench.py 3125 5000 > "$OUT/r2_gemm_mid.json" 2>/dev/null; cat "$OUT/r2_gemm_mid.json"
timeout 600 python -m pytest tests -m gpu -x -q -k "bf16_mode or teacher or clip or full_size or batched_decode or compaction" > "$OUT/r2_pytest_sel.txt" 2>&1
tail -12 "$OUT/r2_pytest_sel.txt"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp --gemm-mode bf16 > "$OUT/r2_bench_greedy_bf16.json" 2> "$OUT/r2_bench_greedy_bf16.err"
python -c "
import json; r=json.load(open('$OUT/r2_bench_greedy_bf16.json')); print('greedy bf16:', r['value'], r['match_vs_fp32'], r['roofline']['achieved'], {k:v['avg_ms'] for k,v in r['kernels'].items()})"
tail -3 "$OUT/r2_bench_greedy_bf16.err"
