#!/bin/bash
# round-2 wrap-up: full GPU suite, RN50x4 numbers, image -> caption side workloads
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r2_pytest_gpu.txt" 2>&1; tail -3 "$OUT/r2_pytest_gpu.txt"
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --clip rn50x4 --captions 2014 > "$OUT/r2_side_image_rn50x4.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --clip rn50x4 --captions 2014 --gemm-mode f16 > "$OUT/r2_side_image_rn50x4_f16.json" 2>/dev/null
timeout 300 python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --workload image_beam --captions 2014 > "$OUT/r2_side_image_f16x2.json" 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2_side_image*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], r["value"], r.get("ms_per_step"), r["config"])
    except Exception as e:
        print(f, "ERR", e)
PY
