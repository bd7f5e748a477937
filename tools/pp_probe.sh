#!/bin/bash
# Correctness (vs fp64) + timing of the forced GEMM geometries on the decode loop's shapes, one process per geometry:
#   gpurun --timeout 900 -- 'bash tools/pp_probe.sh "0 2 10 11 12 13" "3125 25000"'
set -u
GEOS=${1:-"0 10 11 12 13"}
MS=${2:-"3125 25000"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
: > "$OUT/pp_probe.txt"
for g in $GEOS; do
    CAPDEC_SPLITK=0 CAPDEC_H2W=$g CAPDEC_HOOK_PACKA=1 timeout 300 python tools/h2w_probe.py $MS >> "$OUT/pp_probe.txt" 2>> "$OUT/pp_probe.err"
done
cat "$OUT/pp_probe.txt"
tail -5 "$OUT/pp_probe.err"
