#!/bin/bash
# Round 6, call 11: the shortened 1024-position context test, and the metric line + 625-caption shard on the final tree
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/ -x -q -m gpu -k "contexts_up_to or long_context or stop_at" --durations=3 2>&1 | tail -7 | cut -c1-160
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/r6_bench.json" 2> "$OUT/r6_bench.err"
python - "$OUT/r6_bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", r["value"], r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"], r["power"], "checks", r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp = r["stop_profile"]; print("stop", sp.get("compaction_on"), sp.get("compaction_off"), sp.get("oracle_check", {}).get("ok"), sp.get("error"), sp["shards_of_8"]["ms_max_over_mean"])
print("T12", r["entry_length_12"]["value"]); print("cpu", r["cpu_baseline"]["value"])
for k, v in (r.get("other_configs") or {}).items(): print(k, v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("error"))
k = r["kernels"]; print({n: (v["avg_ms"], v.get("tflops")) for n, v in k.items() if n in ("gemm_f16x2p", "gemm_f16x2p_lmhead_topk", "attn_decode")})
PY
timeout 400 python bench.py --cpu-captions 0 --no-checks --cpu-seconds 0 --captions 625 --steps 20 --warmup 5 > "$OUT/r6_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
