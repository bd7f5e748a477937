#!/bin/bash
# the whole GPU suite + smoke() on the final tree (what the driver runs at round end)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/r5_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/r5_pytest_gpu.txt"; tail -34 "$OUT/r5_pytest_gpu.txt" | cut -c1-160
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 5 2>/dev/null | cut -c1-400
