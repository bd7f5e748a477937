#!/bin/bash
# Round 6, call 8 (final tree: train step split into four translation units, first poll never backs off, the
# prefill-attention-forms test): the whole GPU suite, smoke(), the metric line, the 625-caption shard, the train step
set -u
TAG=r6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/${TAG}_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/${TAG}_pytest_gpu.txt"; tail -5 "$OUT/${TAG}_pytest_gpu.txt" | cut -c1-160
cp "$OUT/parity_counts.txt" "$OUT/${TAG}_parity_counts.txt" 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$OUT/${TAG}_pytest_gpu.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python - "$OUT/${TAG}_bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", r["value"], r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["power"], "checks", r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp = r["stop_profile"]; print("stop", sp.get("compaction_on"), sp.get("compaction_off"), sp.get("oracle_check", {}).get("ok"), sp.get("error"), sp["shards_of_8"]["ms_max_over_mean"], sp["captions_identical_on_vs_off"])
print("launched", sp["rows_launched_per_step"][:12])
print("T12", r["entry_length_12"]); print("cpu", r["cpu_baseline"]["value"])
for k, v in (r.get("other_configs") or {}).items(): print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"))
k = r["kernels"]; print({n: v["avg_ms"] for n, v in k.items() if n in ("gemm_f16x2p", "gemm_f16x2p_lmhead_topk", "attn_decode", "layernorm")})
PY
tail -2 "$OUT/${TAG}_bench.err"
B="timeout 400 python bench.py --cpu-captions 0 --no-checks --cpu-seconds 0"
$B --captions 625 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
$B --steps 5 --warmup 2 > "$OUT/tmp5000.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp5000.json'));print('5000 (5 steps, after the 625 run):',r['value'],r['ms_per_step'])"
timeout 200 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_train_bench_prefix.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_train_bench_prefix.json'));print('train prefix:',r['value'],r['ms_per_step'])"
timeout 200 python bench.py --workload train_step --train-scope full --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_train_bench_full.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_train_bench_full.json'));print('train full:',r['value'],r['ms_per_step'])"
