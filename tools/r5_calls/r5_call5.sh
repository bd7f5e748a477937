#!/bin/bash
# round 5, call 5: the one-round-trip greedy attention on the bf16 KV cache -- parity, then A/B against the beam kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "bf16_mode or long_context or finished_caption" --durations=8 2>&1 | tail -16 | tee "$OUT/r5_att_g16_tests.txt"
tail -3 "$OUT/parity_counts.txt"
summ() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm_x1' in n or 'attn_dec' in n or 'layernorm' in n or 'select' in n or 'embed' in n})"; }
for kv in "CAPDEC_ATT_G16=1" "CAPDEC_ATT_G16=0"; do
    echo "-- greedy_mlp bf16 $kv"
    env $kv timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_$(echo $kv | tr ' =' '__').json" | summ
done
echo "-- greedy_mlp bf16, 625 captions"
timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --captions 625 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_625.json" | summ
