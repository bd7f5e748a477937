#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
echo "old bf16p"; CAPDEC_GEMM_MODE=bf16 CAPDEC_X1_OLD=1 timeout 120 python tools/gemm_bench.py 25000 5000 2>/dev/null
echo "new x1 ns4"; CAPDEC_GEMM_MODE=bf16 timeout 120 python tools/gemm_bench.py 25000 5000 2>/dev/null
echo "new x1 ns3"; CAPDEC_GEMM_MODE=bf16 CAPDEC_X1_NS=3 timeout 120 python tools/gemm_bench.py 25000 5000 2>/dev/null
echo "new x1 f16 ns4"; CAPDEC_GEMM_MODE=f16 timeout 120 python tools/gemm_bench.py 25000 2>/dev/null
unset CAPDEC_HOOK_PACKA CAPDEC_HOOK_CACHE
timeout 600 python -m pytest tests -m gpu -x -q -k "bf16_mode or clip_fp16 or teacher" 2>&1 | tail -4
for v in 4 3; do
CAPDEC_X1_NS=$v timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp --gemm-mode bf16 > "$OUT/r2_bench_greedy_bf16_ns$v.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_bench_greedy_bf16_ns$v.json')); print('greedy bf16 ns$v:', r['value'], r['match_vs_fp32'], {k:v['avg_ms'] for k,v in r['kernels'].items() if 'gemm' in k or 'attn_decode' in k})"
done
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --gemm-mode bf16 > "$OUT/r2_bench_beam_bf16.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_bench_beam_bf16.json')); print('beam bf16:', r['value'], {k:(v['avg_ms'], v.get('tflops')) for k,v in r['kernels'].items() if 'gemm' in k or 'attn_decode' in k})"
for m in f16x2 f16; do
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload text_embed --captions 20000 --gemm-mode $m > "$OUT/r2_bench_text_$m.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_bench_text_$m.json')); print('text_embed $m:', r['value'])"
done
