#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or full_size or compaction or make_preds or prompt or bf16_mode" 2>&1 | tail -3
for na in 2 4; do for occ in 4 3; do
CAPDEC_ATT_NA=$na CAPDEC_ATT_OCC=$occ timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_att_na${na}_occ$occ.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_att_na${na}_occ$occ.json')); print('NA=$na OCC=$occ:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
done; done
timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_att_625.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_att_625.json')); print('625:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
