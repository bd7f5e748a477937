#!/bin/bash
# Round 6, call 2 (and 3, after the matrix-core rewrite): the prefill attention for 24..128 positions (CLIP towers, long prefixes) -- parity tests that run it, the
# tower lines after the change, a kernel table of the text tower; the GPT-2-small train parity case; knob A/Bs for
# review items 1a / 3a (split-K of the N = 768 projections off); the reworked stop profile.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
rm -f "$OUT/parity_counts.txt"
timeout 900 python -m pytest tests/ -x -q -m gpu -k "clip or logits or long_context or text_to_prefix or config4 or make_preds_from or gpt2_small_geometry or mapper or bf16_mode_logits" --durations=8 2>&1 | tail -16 | cut -c1-180
cat "$OUT/parity_counts.txt" 2>/dev/null | tail -3
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
for m in f16 f16x2; do
  $B --workload text_embed --captions 20000 --gemm-mode $m --steps 3 --warmup 1 > "$OUT/r6b_text_$m.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6b_text_$m.json'));print('text $m:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
done
$B --workload image_beam --captions 2014 --steps 2 --warmup 1 > "$OUT/r6b_image_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6b_image_f16x2.json'));print('image:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
rm -rf "$OUT/tracetext"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tracetext" -- python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
python tools/trace_summary.py "$OUT/tracetext" "$OUT/r6_clip_text_f16_kernels.txt" --title "bench.py --workload text_embed --captions 20000 --gemm-mode f16 under rocprofv3 --kernel-trace (lane-per-query attention)"; head -12 "$OUT/r6_clip_text_f16_kernels.txt" | cut -c1-50,95-200
rm -rf "$OUT/tracetext"
# knob A/Bs, same box
echo "--- 625 captions: default vs CAPDEC_SPLITK_MID=0"
for e in "X=1" "CAPDEC_SPLITK_MID=0" "X=1" "CAPDEC_SPLITK_MID=0"; do
  env $e $B --captions 625 --steps 10 --warmup 3 --no-smi > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));k=r['kernels'];print('$e',r['value'],r['ms_per_step'],'gemm',k['gemm_f16x2p']['ms_est'],'ln',k['layernorm']['ms_est'])"
done
echo "--- greedy bf16 5000: default vs CAPDEC_X1_SPLITK=0"
for e in "X=1" "CAPDEC_X1_SPLITK=0" "X=1" "CAPDEC_X1_SPLITK=0"; do
  env $e $B --workload greedy_mlp --gemm-mode bf16 --steps 10 --warmup 3 --no-smi > "$OUT/tmp.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp.json'));k=r['kernels'];print('$e',r['value'],r['ms_per_step'],'gemm',k['gemm_x1']['ms_est'],'ln',k['layernorm']['ms_est'])"
done
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > "$OUT/r6b_bench.json" 2> "$OUT/r6b_bench.err"; python - <<PY
import json
r=json.loads([l for l in open("$OUT/r6b_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"])
sp=r["stop_profile"]
for k,v in sp.items():
    if k not in ("rows_launched_per_step","rows_alive_per_step","note"): print(k, json.dumps(v)[:600])
print(r["entry_length_12"])
PY
tail -3 "$OUT/r6b_bench.err"
