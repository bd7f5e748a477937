#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; TAG=r2
cd /tmp && export TMPDIR=/tmp
cd "$R"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/${TAG}_pmc_$c" -- python bench.py --cpu-seconds 0 --steps 1 --warmup 0 \
        > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
python tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_traffic.json" \
    "python bench.py --cpu-seconds 0 --steps 1 --warmup 0 (default workload: 5000 captions, beam 5, T=67)" f16x2 5000
find "$OUT" -name "*counter_collection.csv" -delete
