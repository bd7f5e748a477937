#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
for occ in 4 3; do
CAPDEC_ATT_OCC=$occ timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_bench_dedupe_occ$occ.json" 2>/dev/null
python - <<PY
import json
r=json.load(open("$OUT/r2_bench_dedupe_occ$occ.json")); k=r["kernels"]
print("occ$occ", r["value"], "attn_decode avg", k["attn_decode"]["avg_ms"], "gemm", k["gemm_f16x2p"]["tflops"], "lmhead", k["gemm_f16x2p_lmhead_topk"]["tflops"])
PY
done
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or full_size or compaction or smoke" > "$OUT/r2_pytest_dec.txt" 2>&1
tail -4 "$OUT/r2_pytest_dec.txt"
