#!/bin/bash
# Round 6: everything the round's numbers come from, one gpurun call:
#   gpurun --timeout 3400 -- 'bash tools/r6_final.sh'
# the whole GPU suite + smoke(); the driver-style metric line (with the stop profile, entry_length 12, the other BASELINE
# configs and the CPU baseline); the 625-caption shard and its per-(kernel, grid) table; the metric command under
# rocprofv3 --kernel-trace --stats; FETCH_SIZE / WRITE_SIZE in their own passes; BASELINE configs[1] / [3] / [4] as lines
# of their own (roofline + cpu_baseline) and kernel stats of the text tower; the train step in both scopes.
set -u
TAG=r6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/${TAG}_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/${TAG}_pytest_gpu.txt"; tail -5 "$OUT/${TAG}_pytest_gpu.txt" | cut -c1-160
cp "$OUT/parity_counts.txt" "$OUT/${TAG}_parity_counts.txt" 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$OUT/${TAG}_pytest_gpu.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python - "$OUT/${TAG}_bench.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", r["value"], r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["traffic"], r["power"], "checks", r["oracle_check"]["ok"], r["ids_check"]["ok"])
sp = r["stop_profile"]; print("stop", sp.get("compaction_on"), sp.get("compaction_off"), sp.get("oracle_check", {}).get("ok"), sp.get("error"))
print("T12", r["entry_length_12"]); print("cpu", r["cpu_baseline"]["value"])
for k, v in (r.get("other_configs") or {}).items(): print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"))
PY
tail -3 "$OUT/${TAG}_bench.err"
B="timeout 400 python bench.py --cpu-captions 0 --no-checks"
$B --cpu-seconds 0 --captions 625 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
rm -rf "$OUT/kt625"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt625" -- python bench.py --cpu-seconds 0 --no-checks --no-smi --captions 625 --steps 1 --warmup 1 > /dev/null 2>&1
python tools/trace_summary.py "$OUT/kt625" "$OUT/${TAG}_625_kernels.txt" --title "bench.py --captions 625 --steps 1 --warmup 1 under rocprofv3 --kernel-trace (round-6 tree: two passes of 67 steps + mapper + prefill)"; rm -rf "$OUT/kt625"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_kt" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
find "$OUT/${TAG}_kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
rm -rf "$OUT/${TAG}_kt"
head -6 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --no-checks --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)" f16x2 5000 "${TAG}"
find "$OUT" -name "*counter_collection.csv" -delete; rm -rf "$OUT/${TAG}_pmc_FETCH_SIZE" "$OUT/${TAG}_pmc_WRITE_SIZE"
python -c "import json;r=json.load(open('$OUT/${TAG}_pmc_traffic.json'));print({k:(v.get('traffic_bytes_per_launch') if isinstance(v,dict) else v) for k,v in r.items()})" 2>&1 | cut -c1-600
# ---- the other BASELINE configs as lines of their own
$B --cpu-seconds 0 --workload greedy_mlp --gemm-mode bf16 --steps 10 --warmup 3 > "$OUT/${TAG}_greedy_bf16_bench.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_greedy_bf16_bench.json'));print('greedy bf16:',r['value'],r['ms_per_step'],r['roofline']['frac'])"
$B --cpu-seconds 0 --gemm-mode bf16 --steps 3 --warmup 1 > "$OUT/${TAG}_beam_bf16_bench.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_beam_bf16_bench.json'));print('beam bf16:',r['value'],r['ms_per_step'])"
$B --cpu-seconds 12 --workload text_embed --captions 20000 --gemm-mode f16 --steps 5 --warmup 2 > "$OUT/${TAG}_text_f16.json" 2>/dev/null; tail -c 1300 "$OUT/${TAG}_text_f16.json"; echo
$B --cpu-seconds 12 --workload text_embed --captions 20000 --steps 5 --warmup 2 > "$OUT/${TAG}_text_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_text_f16x2.json'));print('text f16x2:',r['value'],r['roofline'])"
$B --cpu-seconds 15 --workload image_beam --captions 2014 --steps 3 --warmup 1 > "$OUT/${TAG}_image_vit.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_image_vit.json'));print('image vit:',r['value'],r['roofline'],r['cpu_baseline'])"
$B --cpu-seconds 0 --workload image_beam --clip rn50x4 --captions 2014 --steps 3 --warmup 1 > "$OUT/${TAG}_image_rn50x4.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_image_rn50x4.json'));print('image rn50x4:',r['value'],r['roofline'])"
rm -rf "$OUT/ktt"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktt" -- python bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --no-smi > /dev/null 2>&1
find "$OUT/ktt" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_clip_text_f16_kernel_stats.csv" \;
python tools/trace_summary.py "$OUT/ktt" "$OUT/${TAG}_clip_text_f16_kernels.txt" --title "bench.py --workload text_embed --captions 20000 --gemm-mode f16 --steps 2 --warmup 1 under rocprofv3 --kernel-trace (round-6 tree)"; rm -rf "$OUT/ktt"
head -5 "$OUT/${TAG}_clip_text_f16_kernels.txt" | cut -c1-50,95-200
timeout 200 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_train_bench_prefix.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_train_bench_prefix.json'));print('train prefix:',r['value'],r['ms_per_step'])"
timeout 200 python bench.py --workload train_step --train-scope full --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_train_bench_full.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/${TAG}_train_bench_full.json'));print('train full:',r['value'],r['ms_per_step'])"
