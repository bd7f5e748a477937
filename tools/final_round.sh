#!/bin/bash
# Everything the round's numbers come from, one gpurun call:  gpurun --timeout 2000 -- 'bash tools/final_round.sh r3'
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/${TAG}_pytest_gpu.txt" 2>&1; tail -3 "$OUT/${TAG}_pytest_gpu.txt"
bash tools/collect_profiles.sh "$TAG" > "$OUT/${TAG}_collect.log" 2>&1; tail -c 1500 "$OUT/${TAG}_collect.log" | head -30
cd "$R"
bash tools/side_workloads.sh "$TAG"
