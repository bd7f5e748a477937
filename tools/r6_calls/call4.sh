#!/bin/bash
# Round 6, call 4: matrix-core prefill attention with every global load issued up front; adaptive poll cadence of the
# decode loop (stop profile); review item 1a stamps -- the N = 768 projections of the 625-caption step split along K
# (ping-pong 256 x 128, S = 3: default), split on 128 x 128 tiles (CAPDEC_PP=0) and UNSPLIT with a separate LayerNorm
# (CAPDEC_PP=0 CAPDEC_SPLITK_MID=0), per-(kernel, grid) tables of each.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
rm -f "$OUT/parity_counts.txt"
timeout 900 python -m pytest tests/ -x -q -m gpu -k "p40 or clip or long_context or config4 or make_preds_from or compact or midsize or text_to_prefix" --durations=6 2>&1 | tail -12 | cut -c1-180
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
for m in f16 f16x2; do
  $B --workload text_embed --captions 20000 --gemm-mode $m --steps 3 --warmup 1 > "$OUT/r6d_text_$m.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6d_text_$m.json'));print('text $m:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
$B --workload image_beam --captions 2014 --steps 2 --warmup 1 > "$OUT/r6d_image_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6d_image_f16x2.json'));print('image:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
rm -rf "$OUT/tracetext"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tracetext" -- python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
python tools/trace_summary.py "$OUT/tracetext" "$OUT/r6_clip_text_f16_kernels.txt" --title "bench.py --workload text_embed --captions 20000 --gemm-mode f16 under rocprofv3 --kernel-trace (matrix-core prefill attention, loads up front)"; head -8 "$OUT/r6_clip_text_f16_kernels.txt" | cut -c1-50,95-200
rm -rf "$OUT/tracetext"
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > "$OUT/r6d_bench.json" 2> "$OUT/r6d_bench.err"; python - <<PY
import json
r=json.loads([l for l in open("$OUT/r6d_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp=r["stop_profile"]
for k in ("stop_logit_offset","mean_len_best_beam","len_percentiles_10_50_90_max","compaction_on","compaction_off","captions_identical_on_vs_off","row_steps_if_every_caption_left_at_its_own_stop"): print(k, sp.get(k))
print(sp["shards_of_8"]["ms_max_over_mean"], sp["shards_of_8"]["whole_node_captions_per_s_if_8_gpus"], sp["oracle_check"].get("ok"), sp["oracle_check"].get("oracle_equal"))
print("launched", sp.get("rows_launched_per_step"))
PY
tail -2 "$OUT/r6d_bench.err"
echo "--- item 1a: 625 captions, the N = 768 projections"
for tag in default pp0 pp0_unsplit; do
  case $tag in default) E="X=1";; pp0) E="CAPDEC_PP=0";; pp0_unsplit) E="CAPDEC_PP=0 CAPDEC_SPLITK_MID=0";; esac
  env $E $B --captions 625 --steps 10 --warmup 3 --no-smi > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));k=r['kernels'];print('$tag',r['value'],r['ms_per_step'],'gemm',k['gemm_f16x2p']['ms_est'],'ln',k['layernorm']['ms_est'])"
  rm -rf "$OUT/trace1a"; env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace1a" -- python bench.py --captions 625 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
  python tools/trace_summary.py "$OUT/trace1a" "$OUT/r6_625_kernels_$tag.txt" --title "bench.py --captions 625 --steps 1 --warmup 1 under rocprofv3 --kernel-trace, $E"; head -9 "$OUT/r6_625_kernels_$tag.txt" | cut -c1-60,95-200
  rm -rf "$OUT/trace1a"
done
