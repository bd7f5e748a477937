#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r2_pytest_all2.txt" 2>&1
tail -15 "$OUT/r2_pytest_all2.txt"
for occ in 4 3; do
  CAPDEC_ATT_OCC=$occ timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_bench_att_occ$occ.json" 2> "$OUT/r2_bench_att_occ$occ.err"
  python - <<PY
import json
r=json.load(open("$OUT/r2_bench_att_occ$occ.json")); k=r["kernels"]
print("occ$occ", r["value"], "attn_decode avg", k["attn_decode"]["avg_ms"], "gemm", k["gemm_f16x2p"]["tflops"], "lmhead", k["gemm_f16x2p_lmhead_topk"]["tflops"])
PY
done
timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_bench_625b.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_bench_625b.json')); print('625:', r['value'], {k:v['avg_ms'] for k,v in r['kernels'].items()})"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp --gemm-mode bf16 > "$OUT/r2_bench_greedy_bf16.json" 2> "$OUT/r2_bench_greedy_bf16.err"
python -c "
import json; r=json.load(open('$OUT/r2_bench_greedy_bf16.json')); print('greedy bf16:', r['value'], r['match_vs_fp32'], r['roofline']['achieved'])"
tail -3 "$OUT/r2_bench_greedy_bf16.err"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp > "$OUT/r2_bench_greedy_f16x2.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_bench_greedy_f16x2.json')); print('greedy f16x2:', r['value'])"
