#!/bin/bash
# round 5, call 4: deep-ring one-plane ping-pong tiles and NA = 4 in the bf16 decode attention (A/B), the new end-to-end tests,
# the full-scope train bench
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
echo "== new end-to-end tests + forced x1 geometries (deep rings)"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "config4_chain or make_preds_from_captions or (test_wide_single and bf16)" --durations=8 2>&1 | tail -20 | tee "$OUT/r5_e2e_tests.txt"
summ() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm_x1' in n or 'attn_dec' in n})"; }
for kv in "CAPDEC_PP_X1=0" "CAPDEC_PP_X1=2" "CAPDEC_PP_X1=1" "CAPDEC_PP_X1=0 CAPDEC_ATT_NA=4"; do
    echo "-- greedy_mlp bf16 $kv"
    env $kv timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_$(echo $kv | tr ' =' '__').json" | summ
done
for kv in "CAPDEC_PP_X1=0" "CAPDEC_PP_X1=3" "CAPDEC_PP_X1=0 CAPDEC_ATT_NA=4"; do
    echo "-- beam_transformer bf16 $kv"
    env $kv timeout 300 python bench.py --gemm-mode bf16 --steps 3 --warmup 1 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_beam_bf16_$(echo $kv | tr ' =' '__').json" | summ
done
echo "== train bench, both scopes"
for sc in prefix full; do
    timeout 200 python bench.py --workload train_step --train-scope $sc --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tee "$OUT/r5_train_bench_$sc.json" | cut -c1-1600
done
