#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd "$R"
for ns in 4 5 3; do
CAPDEC_H2_NS=$ns timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_625_ns$ns.json" 2>/dev/null
python -c "
import json; r=json.load(open('$OUT/r2_625_ns$ns.json')); print('625 ns$ns:', r['value'], {k:v['avg_ms'] for k,v in r['kernels'].items() if 'gemm' in k})"
done
