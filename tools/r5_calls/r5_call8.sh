#!/bin/bash
# round 5, call 8: K / V through the qkv epilogue in bf16 mode (bf16 cache) -- parity, then A/B with CAPDEC_KV_DIRECT
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "bf16_mode or lm_head_three or finished_caption" --durations=6 2>&1 | tail -12 | tee "$OUT/r5_bf16_kvdirect_tests.txt"
tail -2 "$OUT/parity_counts.txt"
summ() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print(d['value'], d['ms_per_step'], {n:(round(v.get('avg_ms',0)*1000,1), v.get('tflops')) for n,v in k.items() if 'gemm_x1' in n or 'attn_dec' in n})"; }
for kv in "CAPDEC_KV_DIRECT=1" "CAPDEC_KV_DIRECT=0"; do
    echo "-- greedy_mlp bf16 $kv"
    env $kv timeout 200 python bench.py --workload greedy_mlp --gemm-mode bf16 --steps 6 --warmup 2 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_greedy_bf16_$(echo $kv | tr ' =' '__').json" | summ
done
for kv in "CAPDEC_KV_DIRECT=1" "CAPDEC_KV_DIRECT=0"; do
    echo "-- beam bf16 $kv"
    env $kv timeout 300 python bench.py --gemm-mode bf16 --steps 3 --warmup 1 --cpu-seconds 0 --no-checks 2>/dev/null | tee "$OUT/r5_beam_bf16_$(echo $kv | tr ' =' '__').json" | summ
done
