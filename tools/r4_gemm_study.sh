#!/bin/bash
# Round-4 GEMM study, one box, one file (profiles/r4_gemm_pp.txt is a copy of gpurun_out/r4_gemm_pp.txt):
#   gpurun --timeout 1200 -- 'bash tools/r4_gemm_study.sh'
# (1) the kernels of rounds 2-3 against the ping-pong geometries on the decode loop's shapes (isolated 20-launch loops);
# (2) ablations of the ping-pong main loop (measurement build: results wrong by design); (3) per-block phase stamps with
# the direct and the LDS-transposed epilogue; (4) time against K at 3125 rows: fixed cost per launch and time per k-step
# of both structures in back-to-back launches; (5) hot vs cold weights.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
F=$OUT/r4_gemm_pp.txt
cd "$R"
P="python tools/h2w_probe.py"
export CAPDEC_HOOK_PACKA=1
{
echo "== (1) geometries (CAPDEC_H2W forced; 0 = round-2 128x128 pair kernel, 2 = round-3 256x128 single accumulator, 10 / 14 / 12 ="
echo "   ping-pong 256x128 / 256x192 / 256x256), no split-K, 20 back-to-back launches per shape, fp32-equivalent TFLOP/s"
for g in 0 2 10 14 12; do CAPDEC_SPLITK=0 CAPDEC_H2W=$g $P 3125 25000 2>/dev/null; done
echo
echo "== (2) ablations of the ping-pong loop, geometry 10 (256x128), measurement build: 0 = full, 1 = no LDS-DMA in the loop,"
echo "   2 = no fragment reads, 3 = no MFMAs, 4 = no barriers, 5 = MFMAs + barriers only, 6 = no epilogue stores, 7 = non-temporal"
echo "   stores, 8 = direct (uncoalesced) epilogue"
for m in 25000 3125; do for a in 0 1 2 3 4 5 6 7 8; do echo "rows $m ablation $a"; CAPDEC_MEASURE_LIB=1 CAPDEC_PP_ABL=$a CAPDEC_SPLITK=0 CAPDEC_H2W=10 $P $m 2>/dev/null | cut -c60-; done; done
echo
echo "== (3) per-block phase stamps (wall clock, 10 ns ticks; isolated launches): direct epilogue (ablation 8) vs LDS-transposed"
for a in 8 0; do echo "-- epilogue: $([ $a = 8 ] && echo direct || echo LDS-transposed)"; 
  CAPDEC_PP_ABL=$a CAPDEC_H2W=12 python tools/pp_stamps.py "25000,2304,768" 2>/dev/null | grep -E "launch|per block" | tail -2    # 256 x 256 tiles
  CAPDEC_PP_ABL=$a CAPDEC_H2W=10 python tools/pp_stamps.py "3125,2304,768;3125,768,3072" 2>/dev/null | grep -E "launch|per block" | tail -4; done
echo
echo "== (4) 3125 rows x 2304 columns, K = 64 .. 1536, back-to-back launches: (ms at K=1536 - ms at K=768) / 48 = time per k-step per CU"
export PROBE_SHAPES="3125,2304,64;3125,2304,256;3125,2304,768;3125,2304,1536"
for g in 0 10; do CAPDEC_SPLITK=0 CAPDEC_H2W=$g $P 1 2>/dev/null | cut -c1-420; done
unset PROBE_SHAPES
echo
echo "== (5) hot (one weight matrix repeated) vs cold (48 different matrices, > Infinity Cache) weights, 3125 rows, with split-K"
for g in 0 10 14; do echo "geometry $g"; CAPDEC_H2W=$g CAPDEC_HOOK_CACHE=1 python tools/gemm_cold.py 3125 2>/dev/null; done
} > "$F" 2>&1
wc -l "$F"; head -12 "$F" | cut -c1-300
