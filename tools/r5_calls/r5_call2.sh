#!/bin/bash
# round 5, call 2: the reworked train step (dropout, un-normalised gradient flow on the f16x2 kernels, lazy loss) + the
# forced-geometry children three at a time
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"; cd "$R"
echo "== train tests"
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "train_step or train_loop" --durations=10 2>&1 | tail -40 | tee "$OUT/r5_train_tests.txt"
echo "== forced geometries, three at a time"
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "test_wide_single_accumulator" --durations=3 2>&1 | tail -15 | tee "$OUT/r5_forced_geo.txt"
echo "== train bench (default = f16x2 backward)"
timeout 120 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tee "$OUT/r5_train_bench_call2.json" | cut -c1-900
