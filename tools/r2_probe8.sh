#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
export CAPDEC_HOOK_PACKA=1 CAPDEC_HOOK_CACHE=1
for a in 0 1 2 3 4; do
  echo "ABL=$a"; CAPDEC_H2_ABL=$a timeout 120 python tools/gemm_bench.py 25000 2>/dev/null
done
unset CAPDEC_HOOK_PACKA CAPDEC_HOOK_CACHE
timeout 600 python -m pytest tests -m gpu -x -q -k "train_step" 2>&1 | tail -3
