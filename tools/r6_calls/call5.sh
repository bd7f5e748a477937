#!/bin/bash
# Round 6, call 5: the whole GPU suite on the tree so far (new: captions that stop at the headline size, GPT-2-small train
# parity, matrix-core prefill attention at three blocks per CU), smoke(), the tower lines and the driver-style metric line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=15 > "$OUT/r6_pytest_gpu_mid.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/r6_pytest_gpu_mid.txt"; tail -24 "$OUT/r6_pytest_gpu_mid.txt" | cut -c1-160
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
$B --workload text_embed --captions 20000 --gemm-mode f16 --steps 3 --warmup 1 > "$OUT/r6e_text_f16.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6e_text_f16.json'));print('text f16:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
$B --workload image_beam --captions 2014 --steps 2 --warmup 1 > "$OUT/r6e_image_f16x2.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6e_image_f16x2.json'));print('image:',r['value'],r['ms_per_step'],{k:(v['avg_ms'],v['ms_est']) for k,v in r['clip_tower_kernels'].items()})"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/r6e_bench.json" 2> "$OUT/r6e_bench.err"; python - <<PY
import json
r=json.loads([l for l in open("$OUT/r6e_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["oracle_check"]["ok"], r["ids_check"]["ok"], r["power"])
sp=r["stop_profile"]; print(sp["compaction_on"], sp["compaction_off"], sp["oracle_check"].get("ok")); print(r["entry_length_12"]); print(r["cpu_baseline"]["value"])
PY
tail -2 "$OUT/r6e_bench.err"
$B --captions 625 --steps 20 --warmup 5 > "$OUT/r6e_bench_625.json" 2>/dev/null; python -c "import json;r=json.load(open('$OUT/r6e_bench_625.json'));print('625:',r['value'],r['ms_per_step'])"
