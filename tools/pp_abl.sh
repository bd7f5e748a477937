#!/bin/bash
# ablations of the ping-pong GEMM main loop (results are wrong by design for ABL != 0): what bounds a half-phase
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
: > "$OUT/pp_abl.txt"
for g in ${1:-"10 12"}; do for a in 0 1 2 3 4 5; do
    echo "geometry $g ablation $a" >> "$OUT/pp_abl.txt"
    CAPDEC_MEASURE_LIB=1 CAPDEC_PP_ABL=$a CAPDEC_SPLITK=0 CAPDEC_H2W=$g CAPDEC_HOOK_PACKA=1 timeout 300 python tools/h2w_probe.py ${2:-25000} >> "$OUT/pp_abl.txt" 2>> "$OUT/pp_abl.err"
done; done
cat "$OUT/pp_abl.txt" | cut -c1-700
