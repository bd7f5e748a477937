#!/bin/bash
# Round 6, call 7: the poll back-off (a poll that finds no finished caption doubles the interval): the stop test, the
# compaction test, the metric line (stop profile, other configs, cpu baseline) and the 625-caption shard, twice each on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/ -x -q -m gpu -k "stop or compact or midsize or large_launch" 2>&1 | tail -3 | cut -c1-160
B="timeout 400 python bench.py --cpu-captions 0 --no-checks --cpu-seconds 0"
for i in 1 2; do
  $B --captions 625 --steps 20 --warmup 5 > "$OUT/r6_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
  $B --steps 5 --warmup 2 > "$OUT/tmp5000.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/tmp5000.json'));print('5000 (5 steps):',r['value'],r['ms_per_step'])"
done
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/r6_bench.json" 2> "$OUT/r6_bench.err"
python - "$OUT/r6_bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", r["value"], r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["traffic"], r["power"], "checks", r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp = r["stop_profile"]; print("stop", sp.get("compaction_on"), sp.get("compaction_off"), sp.get("oracle_check", {}).get("ok"), sp.get("error"), sp["shards_of_8"]["ms_max_over_mean"], sp["shards_of_8"]["whole_node_captions_per_s_if_8_gpus"])
print("launched", sp["rows_launched_per_step"][:24])
print("T12", r["entry_length_12"]); print("cpu", r["cpu_baseline"]["value"])
for k, v in (r.get("other_configs") or {}).items(): print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"))
PY
tail -2 "$OUT/r6_bench.err"
