#!/bin/bash
# round 5, call 6: block-form attention of the train step -- parity (default + fallback knobs in the child test), bench, per-kernel table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "train_step or train_loop" --durations=6 2>&1 | tail -14 | tee "$OUT/r5_train_tests_blk.txt"
bash tools/train_profile.sh r5b 2>&1 | tail -48 | cut -c1-230
timeout 120 python bench.py --workload train_step --train-scope full --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tee "$OUT/r5b_train_bench_full.json" | cut -c1-600
