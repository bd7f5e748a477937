#!/bin/bash
# Round 6, call 13: single-plane (fp16) operands for the towers' attention in their 16-bit precision modes: the tower tests,
# then the text tower line (fp16) and, unchanged code path, the fp32-accurate one and the image tower
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -x -q -m gpu -k "clip or text_to_prefix or config4 or make_preds_from or prefill_attention_form" 2>&1 | tail -12 | cut -c1-200
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
for i in 1 2; do
$B --workload text_embed --captions 20000 --gemm-mode f16 --steps 3 --warmup 1 > "$OUT/r6f_text_f16.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6f_text_f16.json'));print('text f16:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
$B --workload text_embed --captions 20000 --steps 3 --warmup 1 > "$OUT/r6f_text_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6f_text_f16x2.json'));print('text f16x2:',r['value'],r['ms_per_step'])"
