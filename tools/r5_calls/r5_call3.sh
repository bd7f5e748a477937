#!/bin/bash
# round 5, call 3: the one-plane (bf16) GEMMs on the ping-pong tiles -- parity first, then the A/B of the planner modes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
echo "== bf16 parity (default planner + forced ping-pong tiles + configs[1] size)"
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "bf16_mode or (test_wide_single and bf16) or (test_wide_single and pingpong256x128) or lm_head_three" --durations=12 2>&1 | tail -40 | tee "$OUT/r5_bf16_parity.txt"
tail -12 "$OUT/parity_counts.txt"
echo "== A/B: greedy bf16 (configs[1]) and beam bf16, planner modes"
for m in 0 1 2; do
    echo "-- greedy_mlp bf16 CAPDEC_PP_X1=$m"
    CAPDEC_PP_X1=$m timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_ppx1_$m.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm' in n or 'attn' in n})"
done
for m in 0 2 3; do
    echo "-- beam_transformer bf16 CAPDEC_PP_X1=$m"
    CAPDEC_PP_X1=$m timeout 300 python bench.py --gemm-mode bf16 --steps 3 --warmup 1 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_beam_bf16_ppx1_$m.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm' in n or 'attn' in n})"
done
