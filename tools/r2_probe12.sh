#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd "$R"
CAPDEC_KV_LAYOUT=pos timeout 600 python -m pytest tests -m gpu -x -q -k "decode_tiny or batched_decode or compaction or bf16_mode" 2>&1 | tail -3
for lay in head pos; do
CAPDEC_KV_LAYOUT=$lay timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_kv_$lay.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_kv_$lay.json')); print('layout $lay:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
done
CAPDEC_KV_LAYOUT=pos timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_kv_pos_625.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_kv_pos_625.json')); print('625 pos:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
