#!/bin/bash
# Round 6, call 15: the text tower computes only the positions up to a chunk's last EOT (captions sorted by length): tower
# tests, then the configs[3] line A/B on one box (CAPDEC_CLIP_TRUNC=0 = all 77 positions)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -x -q -m gpu -k "clip or text_to_prefix or config4 or make_preds_from or prefill_attention_form" 2>&1 | tail -12 | cut -c1-220
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
{
for e in "X=1" "CAPDEC_CLIP_TRUNC=0" "X=1" "CAPDEC_CLIP_TRUNC=0"; do
env $e $B --workload text_embed --captions 20000 --gemm-mode f16 --steps 3 --warmup 1 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));print('text f16 $e:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
for e in "X=1" "CAPDEC_CLIP_TRUNC=0"; do
env $e $B --workload text_embed --captions 20000 --steps 3 --warmup 1 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));print('text f16x2 $e:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
} | tee "$OUT/r6_clip_trunc_ab.txt"
