#!/bin/bash
# Round 6, call 17: QuickGELU epilogue with rcp + mul instead of an IEEE division: tower tests, text / image tower lines
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -x -q -m gpu -k "clip or text_to_prefix or config4 or make_preds_from" 2>&1 | tail -4 | cut -c1-220
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
for i in 1 2; do
$B --workload text_embed --captions 20000 --gemm-mode f16 --steps 5 --warmup 2 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));print('text f16:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
$B --workload text_embed --captions 20000 --steps 5 --warmup 2 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));print('text f16x2:',r['value'],r['ms_per_step'])"
$B --workload image_beam --captions 2014 --steps 2 --warmup 1 > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));print('image:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
