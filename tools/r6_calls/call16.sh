#!/bin/bash
# Round 6, call 16 (final tree: text tower truncated at the EOT, towers on the matrix-core attention at every length): the whole GPU suite + smoke(), the metric line, the 625-caption
# shard, the configs[3] lines with their cpu_baseline and the text tower's kernel table / stats
set -u
TAG=r6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
cd /tmp && export TMPDIR=/tmp && cd "$R"
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/${TAG}_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/${TAG}_pytest_gpu.txt"; tail -4 "$OUT/${TAG}_pytest_gpu.txt" | cut -c1-160
cp "$OUT/parity_counts.txt" "$OUT/${TAG}_parity_counts.txt" 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$OUT/${TAG}_pytest_gpu.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python - "$OUT/${TAG}_bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", r["value"], r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"], r["power"], "checks", r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp = r["stop_profile"]; print("stop", sp.get("compaction_on"), sp.get("compaction_off"), sp.get("oracle_check", {}).get("ok"), sp.get("error"), sp["shards_of_8"]["ms_max_over_mean"])
print("T12", r["entry_length_12"]["value"]); print("cpu", r["cpu_baseline"]["value"])
for k, v in (r.get("other_configs") or {}).items(): print(k, v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("error"))
k = r["kernels"]; print({n: (v["avg_ms"], v.get("tflops")) for n, v in k.items() if n in ("gemm_f16x2p", "gemm_f16x2p_lmhead_topk", "attn_decode")})
PY
B="timeout 400 python bench.py --cpu-captions 0 --no-checks"
$B --cpu-seconds 0 --captions 625 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
$B --cpu-seconds 12 --workload text_embed --captions 20000 --gemm-mode f16 --steps 5 --warmup 2 > "$OUT/${TAG}_text_f16.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_text_f16.json'));print('text f16:',r['value'],r['ms_per_step'],r['roofline']['frac'],r['roofline']['share_of_tower'],r['cpu_baseline']['value'],{k:v['avg_ms'] for k,v in r['clip_tower_kernels'].items()})"
$B --cpu-seconds 0 --workload text_embed --captions 20000 --gemm-mode bf16 --steps 3 --warmup 1 > "$OUT/${TAG}_text_bf16.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_text_bf16.json'));print('text bf16:',r['value'],r['ms_per_step'])"
$B --cpu-seconds 12 --workload text_embed --captions 20000 --steps 5 --warmup 2 > "$OUT/${TAG}_text_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_text_f16x2.json'));print('text f16x2:',r['value'],r['ms_per_step'],r['roofline']['frac'],{k:v['avg_ms'] for k,v in r['clip_tower_kernels'].items()})"
$B --cpu-seconds 0 --workload text_embed --captions 100000 --gemm-mode f16 --steps 3 --warmup 1 > "$OUT/${TAG}_text_f16_100k.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_text_f16_100k.json'));print('text f16, 100 000 captions per step:',r['value'],r['ms_per_step'])"
rm -rf "$OUT/ktt"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktt" -- python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
find "$OUT/ktt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_clip_text_f16_kernel_stats.csv" \;
python tools/trace_summary.py "$OUT/ktt" "$OUT/${TAG}_clip_text_f16_kernels.txt" --title "bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 2 --warmup 1 under rocprofv3 --kernel-trace (final tree: positions up to each chunk's last EOT, one fp16 plane per attention operand)"; rm -rf "$OUT/ktt"
head -6 "$OUT/${TAG}_clip_text_f16_kernels.txt" | cut -c1-50,95-200
