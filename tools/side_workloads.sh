#!/bin/bash
# Side workloads of BASELINE.json (configs[1], [3], [4]) and variants of the headline, one MI355X box:
#   gpurun --timeout 1200 -- 'bash tools/side_workloads.sh r3'   ->  gpurun_out/<tag>_side_workloads.json
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
B="timeout 300 python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks --steps 2 --warmup 1"
$B --entry-length 12 > "$OUT/${TAG}_side_T12.json" 2>/dev/null
$B --workload greedy_mlp > "$OUT/${TAG}_side_greedy_f16x2.json" 2>/dev/null
$B --workload greedy_mlp --gemm-mode bf16 > "$OUT/${TAG}_side_greedy_bf16.json" 2>/dev/null
$B --gemm-mode bf16 > "$OUT/${TAG}_side_beam_bf16.json" 2>/dev/null
$B --gemm-mode bf16x3 > "$OUT/${TAG}_side_bf16x3.json" 2>/dev/null
$B --workload text_embed --captions 20000 > "$OUT/${TAG}_side_text_f16x2.json" 2>/dev/null
$B --workload text_embed --captions 20000 --gemm-mode f16 > "$OUT/${TAG}_side_text_f16.json" 2>/dev/null
$B --workload image_beam --captions 2014 > "$OUT/${TAG}_side_image_f16x2.json" 2>/dev/null
$B --workload image_beam --clip rn50x4 --captions 2014 > "$OUT/${TAG}_side_image_rn50x4.json" 2>/dev/null
timeout 120 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_side_train_step.json" 2>/dev/null
timeout 120 python bench.py --workload train_step --train-scope full --steps 10 --warmup 2 --cpu-seconds 5 > "$OUT/${TAG}_side_train_step_full.json" 2>/dev/null
CAPDEC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
   bench.py --gpus 1 --cpu-seconds 0 --cpu-captions 0 --no-checks --steps 2 --warmup 1 > "$OUT/${TAG}_side_dist1.json" 2>/dev/null
python - "$OUT" "$TAG" <<'PY'
import json, glob, sys, os
out, tag = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(f"{out}/{tag}_side_*.json")):
    name = os.path.basename(f)[len(tag) + 6:-5]
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        res[name] = r
        print(name, r["value"], r.get("unit"), r.get("ms_per_step"), r.get("capi_collective"))
    except Exception as e:
        print(name, "ERR", e)
json.dump(res, open(f"{out}/{tag}_side_workloads.json", "w"), indent=1)
PY
