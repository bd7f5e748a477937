#!/bin/bash
# the whole GPU suite with durations (what the driver runs at round end: pytest tests/ -x -q -m gpu), then smoke()
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
rm -f "$OUT/parity_counts.txt"
SECONDS=0; timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=25 > "$OUT/r5_pytest_gpu.txt" 2>&1
echo "suite wall seconds: $SECONDS" | tee -a "$OUT/r5_pytest_gpu.txt"; tail -42 "$OUT/r5_pytest_gpu.txt" | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
