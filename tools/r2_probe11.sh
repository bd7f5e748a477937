#!/bin/bash
# (historical: the CAPDEC_ATT_SPLIT_MAX variant this probe compared was measured slower at every size and removed; the
#  script still prints the batch-size curve 625 / 1250 / 2500 captions quoted in profiles/r2_gemm_ablations.txt)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or full_size or compaction or attention_split or make_preds or prompt" 2>&1 | tail -4
for mx in 0 16384; do
CAPDEC_ATT_SPLIT_MAX=$mx timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_625_split$mx.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_625_split$mx.json')); print('625 split_max=$mx:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
done
for n in 1250 2500; do
for mx in 0 1000000; do
CAPDEC_ATT_SPLIT_MAX=$mx timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --captions $n > "$OUT/r2_${n}_split$mx.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_${n}_split$mx.json')); print('$n split_max=$mx:', r['value'], 'attn', r['kernels']['attn_decode']['avg_ms'])"
done; done
