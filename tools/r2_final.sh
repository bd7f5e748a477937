#!/bin/bash
# round-2 final measurement batch: full GPU suite, profiles (bench + rocprofv3 kernel stats + PMC), side workloads
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/r2_pytest_final.txt" 2>&1
tail -6 "$OUT/r2_pytest_final.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/r2_smoke.txt" 2>&1; tail -2 "$OUT/r2_smoke.txt"
bash tools/collect_profiles.sh r2 > "$OUT/r2_collect.log" 2>&1
tail -30 "$OUT/r2_collect.log"
cd "$R"
timeout 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --captions 625 > "$OUT/r2_side_625.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --gemm-mode bf16x3 > "$OUT/r2_side_bf16x3.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --entry-length 12 > "$OUT/r2_side_T12.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp > "$OUT/r2_side_greedy_f16x2.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload greedy_mlp --gemm-mode bf16 > "$OUT/r2_side_greedy_bf16.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --gemm-mode bf16 > "$OUT/r2_side_beam_bf16.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload text_embed --captions 20000 > "$OUT/r2_side_text_f16x2.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload text_embed --captions 20000 --gemm-mode f16 > "$OUT/r2_side_text_f16.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --captions 2014 > "$OUT/r2_side_image_f16x2.json" 2>/dev/null
CAPDEC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
   bench.py --gpus 1 --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_side_dist1.json" 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2_side_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], r["value"], r.get("ms_per_step"), (r.get("roofline") or {}).get("achieved"), r.get("capi_collective"), r.get("match_vs_fp32"))
    except Exception as e:
        print(f, "ERR", e)
PY
