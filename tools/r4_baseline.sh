#!/bin/bash
# Same-box baseline of the two regimes + a per-kernel trace of the 625-caption loop:  gpurun --timeout 900 -- 'bash tools/r4_baseline.sh r4a'
set -u
TAG=${1:-r4a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
B="python bench.py --cpu-seconds 0 --cpu-captions 0 --no-checks"
timeout 300 $B --steps 3 --warmup 1 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
timeout 200 $B --captions 625 --steps 10 --warmup 2 > "$OUT/${TAG}_bench_625.json" 2> "$OUT/${TAG}_bench_625.err"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_kt625" -- $B --no-smi --captions 625 --steps 1 --warmup 1 \
    > "$OUT/${TAG}_kt625.json" 2> "$OUT/${TAG}_kt625.err"
python tools/trace_summary.py "$OUT/${TAG}_kt625" "$OUT/${TAG}_625_kernels.txt" --title "bench.py --captions 625 --steps 1 --warmup 1 under rocprofv3 --kernel-trace (two passes of 67 steps + mapper + prefill)"
find "$OUT/${TAG}_kt625" -name "*.csv" -delete
python - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
for n in ("bench", "bench_625"):
    try:
        r = json.loads([l for l in open(f"{out}/{tag}_{n}.json") if l.startswith("{")][-1])
        print(n, r["value"], r["ms_per_step"], r["roofline"]["achieved"], r.get("power"))
    except Exception as e:
        print(n, "ERR", e)
PY
head -40 "$OUT/${TAG}_625_kernels.txt"
