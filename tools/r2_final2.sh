#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r2_pytest_gpu.txt" 2>&1; tail -3 "$OUT/r2_pytest_gpu.txt"
bash tools/collect_profiles.sh r2 > "$OUT/r2_collect.log" 2>&1
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
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --clip rn50x4 --captions 2014 > "$OUT/r2_side_image_rn50x4.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --clip rn50x4 --captions 2014 --gemm-mode f16 > "$OUT/r2_side_image_rn50x4_f16.json" 2>/dev/null
bash tools/r2_rn50.sh > "$OUT/r2_rn50.log" 2>&1
CAPDEC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
   bench.py --gpus 1 --cpu-seconds 0 --steps 2 --warmup 1 > "$OUT/r2_side_dist1.json" 2>/dev/null
python - <<'PY'
import json, glob
r = json.load(open("gpurun_out/r2_bench.json")); print("HEADLINE", r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["roofline"]["traffic"], {k: v["avg_ms"] for k, v in r["kernels"].items()})
for f in sorted(glob.glob("gpurun_out/r2_side_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], r["value"], r.get("ms_per_step"), (r.get("roofline") or {}).get("achieved"), r.get("capi_collective"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -5 "$OUT/r2_bench_kernel_stats.csv" | cut -c1-150
