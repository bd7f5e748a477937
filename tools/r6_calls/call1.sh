#!/bin/bash
# Round 6, first call: the round-5 kernels on a round-6 box (same-box baselines of everything this round changes) and the
# new untimed extras of the metric line (stop profile, entry_length 12).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
cd /tmp && export TMPDIR=/tmp && cd "$R"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
timeout 600 python bench.py --steps 3 --warmup 1 > "$OUT/r6a_bench.json" 2> "$OUT/r6a_bench.err"; tail -c 2500 "$OUT/r6a_bench.json"; echo; tail -5 "$OUT/r6a_bench.err"
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
$B --captions 625 --steps 20 --warmup 5 > "$OUT/r6a_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6a_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
$B --workload text_embed --captions 20000 --gemm-mode f16 --steps 3 --warmup 1 > "$OUT/r6a_text_f16.json" 2>/dev/null; tail -c 1500 "$OUT/r6a_text_f16.json"; echo
$B --workload text_embed --captions 20000 --steps 3 --warmup 1 > "$OUT/r6a_text_f16x2.json" 2>/dev/null; tail -c 600 "$OUT/r6a_text_f16x2.json"; echo
$B --workload image_beam --captions 2014 --steps 2 --warmup 1 > "$OUT/r6a_image_f16x2.json" 2>/dev/null; tail -c 600 "$OUT/r6a_image_f16x2.json"; echo
$B --workload greedy_mlp --gemm-mode bf16 --steps 10 --warmup 3 > "$OUT/r6a_greedy_bf16.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6a_greedy_bf16.json'));print('greedy bf16:',r['value'],r['ms_per_step'])"
# per-(kernel, grid) table of the 625-caption decode loop BEFORE this round's changes
rm -rf "$OUT/trace625"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace625" -- python bench.py --captions 625 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
python tools/trace_summary.py "$OUT/trace625" "$OUT/r6_625_kernels_before.txt" --title "bench.py --captions 625 --steps 1 --warmup 1 under rocprofv3 --kernel-trace (round-5 kernels, round-6 box)"; head -24 "$OUT/r6_625_kernels_before.txt" | cut -c60-250
rm -rf "$OUT/trace625"
rm -rf "$OUT/tracetext"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tracetext" -- python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
python tools/trace_summary.py "$OUT/tracetext" "$OUT/r6_clip_text_f16_kernels_before.txt" --title "bench.py --workload text_embed --captions 20000 --gemm-mode f16 under rocprofv3 --kernel-trace (round-5 kernels)"; head -16 "$OUT/r6_clip_text_f16_kernels_before.txt" | cut -c60-250
rm -rf "$OUT/tracetext"
